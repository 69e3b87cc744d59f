// Test infrastructure: C entry points over the REFERENCE's own kernel bodies, compiled for the host.
// build_ref.sh extracts the kernel part (everything before the ATen launchers) of
//   /root/reference/wisp/csrc/ops/hashgrid_interpolate_cuda.cu  and  .../uniform_sample_cuda.cu
// into oracle/_ref/*.inc (git-ignored; reference sources are never committed) and compiles this file against
// them and against /root/reference/wisp/csrc/ops/hash_utils.cuh where it lies.
#include <ATen/ATen.h>
#include <cmath>
// tap on the cell computation of the 3-D kernels (armed by ref_cells_3d only): x = the clamped scaled position (float3),
// pos = its floor (int3), exactly as the reference kernel holds them after hashgrid_interpolate_cuda.cu:40-43
static thread_local float* g_tap_x = nullptr;
static thread_local int32_t* g_tap_pos = nullptr;
#define REF_TAP3(i, x, pos)                                                                          \
    if (g_tap_x) {                                                                                   \
        g_tap_x[3 * (i)] = (x).x; g_tap_x[3 * (i) + 1] = (x).y; g_tap_x[3 * (i) + 2] = (x).z;      \
        g_tap_pos[3 * (i)] = (pos).x; g_tap_pos[3 * (i) + 1] = (pos).y; g_tap_pos[3 * (i) + 2] = (pos).z; \
    }
#include "_ref/hashgrid_kernels.inc"
}   // closes `namespace wisp` left open by the extracted fragment
#include "_ref/uniform_kernels.inc"
}   // closes `namespace wisp`
#include "_ref/depth_bound_kernel.inc"
}   // closes `namespace wisp`
#include "_ref/query_fwd_kernel.inc"
#include "_ref/query_bwd_kernel.inc"
}   // closes `namespace wisp` (opened by the forward fragment)

extern "C" {

int32_t ref_hash_index_3d(int x, int y, int z, int32_t resolution, int32_t codebook_size) {
    return wisp::hash_index_3d(make_int3(x, y, z), resolution, codebook_size);
}
int32_t ref_hash_index_2d(int x, int y, int32_t resolution, int32_t codebook_size) {
    return wisp::hash_index_2d(make_int2(x, y), resolution, resolution, codebook_size);
}
float ref_clamp(float x, float a, float b) { return wisp::clamp(x, a, b); }

// The cell arithmetic of the reference's 3-D interpolation kernel on `n` coordinates: its own code runs (feature_dim = 0, so the
// blend loop has no iterations and no table is read); the tap reports x [n,3] and pos [n,3].
void ref_cells_3d(int64_t n, int32_t resolution, int32_t codebook_size, const float* coords, float* x_out, int32_t* pos_out) {
    const int64_t first_idx[2] = {0, 0};
    g_tap_x = x_out; g_tap_pos = pos_out;
    wisp::hashgrid_interpolate_3d_cuda_kernel<float>(n, codebook_size, 0, resolution, 0, 1, coords, (const float*)nullptr,
                                                     first_idx, (float*)nullptr);
    g_tap_x = nullptr; g_tap_pos = nullptr;
}

// Host evaluation of THIS package's device formula (csrc/hashgrid.hip corner_setup): ONE fp32 fma per axis instead of the
// reference's double expression.  Not reference code: it is here because the correctly rounded fmaf lives in libm, and the
// CPU suite compares it with ref_cells_3d on adversarial coordinates; the GPU suite compares the device with both.
void fma_cells(int64_t n, int32_t resolution, const float* coords, float* x_out, int32_t* pos_out) {
    const float hi = (float)((double)(resolution - 1) - 1e-5);
    const float hr = 0.5f * (float)resolution;
    for (int64_t i = 0; i < n; ++i) {
        float x = fmaf(hr, coords[i], hr);
        x = fmaxf(0.0f, fminf(hi, x));
        x_out[i] = x;
        pos_out[i] = (int32_t)floorf(x);
    }
}

// one level, float tables: mirrors hashgrid_interpolate_cuda_impl's per-level launch (hashgrid_interpolate_cuda.cu:341-390)
void ref_hashgrid_fwd_level(int64_t n, int32_t codebook_size, int64_t feature_dim, int32_t resolution, int32_t lod_idx,
                            int32_t num_lods, int coord_dim, const float* coords, const float* codebook,
                            const int64_t* first_idx, float* feats) {
    if (coord_dim == 3)
        wisp::hashgrid_interpolate_3d_cuda_kernel<float>(n, codebook_size, feature_dim, resolution, lod_idx, num_lods,
                                                         coords, codebook, first_idx, feats);
    else
        wisp::hashgrid_interpolate_2d_cuda_kernel<float>(n, codebook_size, feature_dim, resolution, lod_idx, num_lods,
                                                         coords, codebook, first_idx, feats);
}

void ref_hashgrid_bwd_level(int64_t n, int32_t codebook_size, int64_t feature_dim, int32_t resolution, int32_t lod_idx,
                            int32_t num_lods, int coord_dim, const float* coords, const float* codebook,
                            const int64_t* first_idx, const float* grad_output, float* grad_codebook) {
    if (coord_dim == 3)
        wisp::hashgrid_interpolate_3d_backward_cuda_kernel<float>(n, codebook_size, feature_dim, resolution, lod_idx,
                                                                  num_lods, false, coords, codebook, first_idx,
                                                                  grad_output, grad_codebook, nullptr);
    else
        wisp::hashgrid_interpolate_2d_backward_cuda_kernel<float>(n, codebook_size, feature_dim, resolution, lod_idx,
                                                                  num_lods, false, coords, codebook, first_idx,
                                                                  grad_output, grad_codebook, nullptr);
}

// the same per-level launch with require_grad_coords = true (hashgrid_interpolate.cpp:88-97): grad_coords [n, 3] is ACCUMULATED
// into by every level's launch, grad_codebook likewise (the caller zeroes both, like the ATen wrapper's at::zeros)
void ref_hashgrid_bwd_level_coords(int64_t n, int32_t codebook_size, int64_t feature_dim, int32_t resolution, int32_t lod_idx,
                                   int32_t num_lods, int coord_dim, const float* coords, const float* codebook,
                                   const int64_t* first_idx, const float* grad_output, float* grad_codebook, float* grad_coords) {
    if (coord_dim == 3)
        wisp::hashgrid_interpolate_3d_backward_cuda_kernel<float>(n, codebook_size, feature_dim, resolution, lod_idx,
                                                                  num_lods, true, coords, codebook, first_idx,
                                                                  grad_output, grad_codebook, grad_coords);
    else
        wisp::hashgrid_interpolate_2d_backward_cuda_kernel<float>(n, codebook_size, feature_dim, resolution, lod_idx,
                                                                  num_lods, true, coords, codebook, first_idx,
                                                                  grad_output, grad_codebook, grad_coords);
}

// hashgrid_query_cuda_kernel / _backward_ (hashgrid_query_cuda.cu:19-66, :100-171), one level, float tables
void ref_hashgrid_query_level(int64_t n, int32_t codebook_size, int32_t probe_size, int64_t feature_dim, int32_t resolution,
                              int32_t lod_idx, int32_t num_lods, const float* coords, const float* codebook, float* feats) {
    wisp::hashgrid_query_cuda_kernel<float>(n, codebook_size, probe_size, feature_dim, resolution, lod_idx, num_lods, coords,
                                            codebook, feats);
}
void ref_hashgrid_query_bwd_level(int64_t n, int32_t codebook_size, int32_t probe_size, int64_t feature_dim, int32_t resolution,
                                  int32_t lod_idx, int32_t num_lods, const float* coords, const float* grad_output,
                                  float* grad_codebook) {
    wisp::hashgrid_query_backward_cuda_kernel<float>(n, codebook_size, probe_size, feature_dim, resolution, lod_idx, num_lods,
                                                     coords, grad_output, grad_codebook);
}

// find_depth_bound_cuda_kernel (render/find_depth_bound_cuda.cu:16-45), one "thread" per pack; curr_idxes_out starts as -1
// like the ATen wrapper's at::zeros_like(...) - 1 (find_depth_bound.cpp:23-36)
void ref_find_depth_bound(int64_t num_packs, int64_t num_nugs, const float* query, const int* curr_in, int* curr_out,
                          const float* depth) {
    for (int64_t t = 0; t < num_packs; ++t) curr_out[t] = -1;
    for (int64_t t = 0; t < num_packs; ++t)
        wisp::find_depth_bound_cuda_kernel_at((wisp::uint)t, num_packs, num_nugs, query, curr_in, curr_out,
                                              reinterpret_cast<const float2*>(depth));
}

// uniform_sample_cuda_kernel (uniform_sample_cuda.cu:18-59) over all nuggets
void ref_uniform_sample(int32_t num_voxels, float scale, const int32_t* ridx, const float* depth, const int32_t* insum,
                        int64_t* new_ridx, float* depth_samples, bool* boundary) {
    for (int32_t t = 0; t < num_voxels; ++t) {
        // the kernel has no grid-stride loop: emulate one thread per voxel by shifting the base pointers
        wisp::uniform_sample_cuda_kernel_at(t, num_voxels, scale, 1.0f / scale, ridx, reinterpret_cast<const float2*>(depth),
                                            insum, new_ridx, depth_samples, boundary);
    }
}
}
