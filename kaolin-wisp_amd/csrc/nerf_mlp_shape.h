// Shape constants of the fused radiance-field decoder build (hidden 64, 4 view octaves, up to 32 grid features: every
// app/nerf config of the reference - 32 for nerf_hash.yaml, 5 for nerf_octree / nerf_codebook, 12 for nerf_triplanar)
// and the packed parameter order  W1[H,IN] b1[H] W2[16,H] b2[16] W3[H,X2] b3[H] W4[H,H] b4[H] W5[3,H] b5[3]
// (wisp/models/nefs/nerf.py:151-173 in nn.Module order).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace wisp_mlp {

constexpr int IN = 32;                    // grid feature width the kernels compute with (narrower inputs are zero-padded)
constexpr int H = 64;                     // hidden width
constexpr int NF = 4;                     // view-direction octaves
constexpr int PE = 3 + 6 * NF;            // 27
constexpr int X2 = 15 + PE;               // 42 real colour-MLP inputs
constexpr int TS = 32;                    // samples per wave tile

constexpr int OW1 = 0, OB1 = OW1 + H * IN, OW2 = OB1 + H, OB2 = OW2 + 16 * H, OW3 = OB2 + 16, OB3 = OW3 + H * X2,
              OW4 = OB3 + H, OB4 = OW4 + H * H, OW5 = OB4 + H, OB5 = OW5 + 3 * H, NPARAM = OB5 + 3;
constexpr int NPARAM_PAD = (NPARAM + 63) / 64 * 64;

// The kernels stage parameters and write gradient partials in the CANONICAL order above (W1 rows IN wide).  The caller's
// packed buffer holds W1 as [H, in_dim]: canonical index -> index in that buffer, or -1 for a padding column of W1.
__host__ __device__ inline int packed_index(int canonical, int in_dim) {
    if (canonical < H * IN) {
        const int r = canonical / IN, c = canonical % IN;
        return c < in_dim ? r * in_dim + c : -1;
    }
    return canonical - H * (IN - in_dim);
}
__device__ __forceinline__ float packed_param(const float* __restrict__ params, int canonical, int in_dim) {
    const int i = packed_index(canonical, in_dim);
    return i < 0 ? 0.0f : params[i];
}

// bf16 matrix-core path (nerf_mlp_bf16.hip); feats/grad_feats element type: 0 f32, 1 f16, 2 bf16 (wisp_hip.h dtype codes).
// Both return 0 or a wisp error code.  `partials` = workspace of wisp_nerf_mlp_workspace_floats() floats.
int bf16_forward(const void* feats, int dtype_io, const float* dirs, int64_t num_samples, int in_dim, const float* params,
                 float* rgb, float* density, hipStream_t st);
int bf16_backward(const void* feats, int dtype_io, const float* dirs, int64_t num_samples, int in_dim, const float* params,
                  const float* grad_rgb, const float* grad_density, void* grad_feats, float* partials, int* partial_rows,
                  hipStream_t st);

// The same kernels with the view direction given per RAY: `code` = [num_rays][32] bf16 from bf16_dir_code (the encoded direction
// of every ray, computed once), `ridx` = ray of every sample.  Every feature width (1..32) and I/O type of the per-sample kernels (bf16_rays_supported).
bool bf16_rays_supported(int dtype_io, int in_dim);
void bf16_dir_code(const float* dirs, int64_t num_rays, void* code, hipStream_t st);
int bf16_forward_rays(const void* feats, int dtype_io, const void* code, const int64_t* ridx, int64_t num_samples, int in_dim,
                      const float* params, float* rgb, float* density, hipStream_t st);
int bf16_backward_rays(const void* feats, int dtype_io, const void* code, const int64_t* ridx, int64_t num_samples, int in_dim,
                       const float* params, const float* grad_rgb, const float* grad_density, void* grad_feats, float* partials,
                       int* partial_rows, hipStream_t st);

// wide decoders (hidden 128; nerf_mlp_wide.hip): bf16 compute only.  `workspace` = wide_workspace_bytes(num_samples, hidden).
bool wide_supported(int hidden);
int64_t wide_workspace_bytes(int64_t num_samples, int hidden);
int wide_forward_dispatch(const void* feats, int dtype_io, const float* dirs, int64_t num_samples, int in_dim, int hidden,
                          const float* params, float* rgb, float* density, hipStream_t st);
int wide_backward_dispatch(const void* feats, int dtype_io, const float* dirs, int64_t num_samples, int in_dim, int hidden,
                           const float* params, const float* grad_rgb, const float* grad_density, void* grad_feats,
                           float* grad_params, void* workspace, hipStream_t st);

}  // namespace wisp_mlp
