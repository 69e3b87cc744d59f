// Trilinear interpolation over the dual octree (SPC "trinkets") for gfx950.
//
// Replaces kaolin.ops.spc.unbatched_interpolate_trilinear (wisp/models/grids/octree_grid.py:147-149) and
// kaolin.ops.spc.coords_to_trilinear_coeffs (wisp/models/grids/codebook_grid.py:164, octree_grid.py:157);
// semantics SURVEY.md Appendix A.6:
//     x = 2^level (0.5 c + 0.5) - point[pidx] ;  coeff_j = prod_axis (bit_j ? x : 1 - x), j = dx<<2 | dy<<1 | dz
//     out = sum_j feats[trinkets[pidx, j]] * coeff_j ;  pidx == -1 -> 0
// Layout: one workgroup row per (voxel, sample); consecutive lanes own consecutive feature channels, so the 8 corner
// rows are read as contiguous segments (C = 16 floats = one 64-B line per corner) and the output row is written
// contiguously.  `half_round` reproduces the reference call `feats.half() ... .float()` (octree_grid.py:147-149)
// without materialising a half copy of the feature tensor: features and the result are rounded through fp16 in
// registers.
#include "wisp_common.h"

static __device__ __forceinline__ void trilinear_coeffs(const float* __restrict__ c, const int16_t* __restrict__ pt,
                                                        int level, float (&w)[8]) {
    const float res = (float)(1 << level);
    float f[3], g[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        f[a] = res * (0.5f * c[a] + 0.5f) - (float)pt[a];
        g[a] = 1.0f - f[a];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = ((j & 4) ? f[0] : g[0]) * ((j & 2) ? f[1] : g[1]) * ((j & 1) ? f[2] : g[2]);
}

__global__ void __launch_bounds__(256)
spc_trilinear_coeffs_kernel(const float* __restrict__ coords, const int16_t* __restrict__ pts, int64_t n, int spv,
                            int level, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (voxel, sample) flattened
    if (i >= n * spv) return;
    const int64_t v = i / spv;
    float w[8];
    trilinear_coeffs(coords + i * 3, pts + v * 3, level, w);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[i * 8 + j] = w[j];
}

template <typename T, typename I>
__global__ void __launch_bounds__(256)
spc_trilinear_fwd_kernel(const float* __restrict__ coords, const I* __restrict__ pidx, const int16_t* __restrict__ points,
                         const int32_t* __restrict__ trinkets, const T* __restrict__ feats, int64_t n, int spv, int channels,
                         int level, int half_round, float* __restrict__ out) {
    const int cpt = channels <= 64 ? channels : 64;                  // lanes cooperating on one row
    const int rows_per_block = blockDim.x / cpt;
    const int ch0 = threadIdx.x % cpt;
    const int64_t stride = (int64_t)gridDim.x * rows_per_block;
    for (int64_t i = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / cpt; i < n * spv; i += stride) {
        if (threadIdx.x / cpt >= rows_per_block) break;
        const int64_t v = i / spv;
        const int64_t p = (int64_t)pidx[v];
        if (p < 0) {
            for (int ch = ch0; ch < channels; ch += cpt) out[i * channels + ch] = 0.0f;
            continue;
        }
        float w[8];
        trilinear_coeffs(coords + i * 3, points + p * 3, level, w);
        const int32_t* tr = trinkets + p * 8;
        for (int ch = ch0; ch < channels; ch += cpt) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float fv = Cvt<T>::to_f(feats[(int64_t)tr[j] * channels + ch]);
                if (half_round) fv = __half2float(__float2half_rn(fv));
                acc += fv * w[j];
            }
            out[i * channels + ch] = half_round ? __half2float(__float2half_rn(acc)) : acc;
        }
    }
}

static inline int interp_grid(int64_t rows, int channels) {
    const int cpt = channels <= 64 ? channels : 64;
    const int rpb = 256 / cpt;
    return (int)min64(ceil_div64(rows, rpb), 16384);
}

extern "C" int wisp_spc_trilinear_coeffs(const float* coords, const int16_t* voxel_points, int64_t num_voxels,
                                         int samples_per_voxel, int level, float* coeffs, wisp_stream_t stream) {
    WISP_REQUIRE(num_voxels >= 0 && samples_per_voxel >= 1 && level >= 0 && level <= 15, "bad sizes");
    if (num_voxels == 0) return WISP_OK;
    WISP_REQUIRE(coords && voxel_points && coeffs, "null pointer");
    const int64_t rows = num_voxels * samples_per_voxel;
    hipLaunchKernelGGL(spc_trilinear_coeffs_kernel, dim3((unsigned)ceil_div64(rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       coords, voxel_points, num_voxels, samples_per_voxel, level, coeffs);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_spc_trilinear_fwd(const float* coords, const void* pidx, int pidx_is_i64, const int16_t* points,
                                      const int32_t* trinkets, const void* feats, int dtype, int64_t num_voxels,
                                      int samples_per_voxel, int channels, int level, int half_round, float* out,
                                      wisp_stream_t stream) {
    WISP_REQUIRE(num_voxels >= 0 && samples_per_voxel >= 1 && channels >= 1 && level >= 0 && level <= 15, "bad sizes");
    if (num_voxels == 0) return WISP_OK;
    WISP_REQUIRE(coords && pidx && points && trinkets && feats && out, "null pointer");
    WISP_REQUIRE(channels <= 64 ? (256 % channels == 0 || true) : true, "channels");
    const int64_t rows = num_voxels * samples_per_voxel;
    const dim3 grid(interp_grid(rows, channels)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define TRI_FWD(T, I)                                                                                              \
    hipLaunchKernelGGL((spc_trilinear_fwd_kernel<T, I>), grid, block, 0, s, coords, (const I*)pidx, points, trinkets, \
                       (const T*)feats, num_voxels, samples_per_voxel, channels, level, half_round, out)
    if (pidx_is_i64) {
        if (dtype == WISP_F32) TRI_FWD(float, int64_t); else if (dtype == WISP_F16) TRI_FWD(__half, int64_t); else TRI_FWD(__hip_bfloat16, int64_t);
    } else {
        if (dtype == WISP_F32) TRI_FWD(float, int32_t); else if (dtype == WISP_F16) TRI_FWD(__half, int32_t); else TRI_FWD(__hip_bfloat16, int32_t);
    }
#undef TRI_FWD
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- all LODs in one launch
// OctreeGrid.interpolate (wisp/models/grids/octree_grid.py:183-219) evaluates one trilinear lookup per active level and
// concatenates or sums the results; here one launch walks the levels for every sample (the level loop is uniform, the
// per-level feature pointer and octree level are scalar loads) and writes the 'cat' row or the 'sum' directly - no
// per-level output tensors, no cat / reshape / sum kernels.  Same arithmetic per level as spc_trilinear_fwd_kernel.
#define SPC_MAX_LODS 16
struct MultiLod { const void* feats[SPC_MAX_LODS]; float* grad[SPC_MAX_LODS]; int32_t level[SPC_MAX_LODS]; };

template <typename T>
__global__ void __launch_bounds__(256)
spc_trilinear_multi_fwd_kernel(const float* __restrict__ coords, const int64_t* __restrict__ chain, int64_t chain_stride,
                               const int16_t* __restrict__ points, const int32_t* __restrict__ trinkets, MultiLod ml,
                               int64_t n, int num_lods, int channels, int half_round, int sum, float* __restrict__ out) {
    const int cpt = channels <= 64 ? channels : 64;
    const int rows_per_block = blockDim.x / cpt;
    const int ch0 = threadIdx.x % cpt;
    const int64_t stride = (int64_t)gridDim.x * rows_per_block;
    const int out_row = sum ? channels : num_lods * channels;
    for (int64_t i = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / cpt; i < n; i += stride) {
        if (threadIdx.x / cpt >= rows_per_block) break;
        for (int ch = ch0; ch < channels; ch += cpt) {
            float total = 0.0f;
            for (int l = 0; l < num_lods; ++l) {
                const int64_t p = chain[i * chain_stride + l];
                float acc = 0.0f;
                if (p >= 0) {
                    float w[8];
                    trilinear_coeffs(coords + i * 3, points + p * 3, ml.level[l], w);
                    const int32_t* tr = trinkets + p * 8;
                    const T* feats = reinterpret_cast<const T*>(ml.feats[l]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float fv = Cvt<T>::to_f(feats[(int64_t)tr[j] * channels + ch]);
                        if (half_round) fv = __half2float(__float2half_rn(fv));
                        acc += fv * w[j];
                    }
                    if (half_round) acc = __half2float(__float2half_rn(acc));
                }
                if (sum) total += acc; else out[i * out_row + l * channels + ch] = acc;
            }
            if (sum) out[i * out_row + ch] = total;
        }
    }
}

static int fill_multi(MultiLod& ml, const void* const* feats, float* const* grads, const int32_t* levels, int num_lods) {
    for (int l = 0; l < SPC_MAX_LODS; ++l) { ml.feats[l] = nullptr; ml.grad[l] = nullptr; ml.level[l] = 0; }
    for (int l = 0; l < num_lods; ++l) {
        if (levels[l] < 0 || levels[l] > 15) return -1;
        ml.level[l] = levels[l];
        if (feats) { if (!feats[l]) return -1; ml.feats[l] = feats[l]; }
        if (grads) { if (!grads[l]) return -1; ml.grad[l] = grads[l]; }
    }
    return 0;
}

// Few channels (nerf_octree / nerf_codebook.yaml: 5): with `channels` lanes per sample every lane repeats the chain and point
// loads and the eight weights, and a 256-thread block holds 51 samples.  Here one thread owns one sample - the weights once, the
// C channels of a corner row from consecutive addresses - with the same products in the same order as the kernel above
// (bit-identical results).  0.158 -> ~0.08 ms for the four levels of the VQAD bench shape (2 M samples).
template <typename T, int C>
__global__ void __launch_bounds__(256)
spc_trilinear_multi_fwd_small_kernel(const float* __restrict__ coords, const int64_t* __restrict__ chain, int64_t chain_stride,
                                     const int16_t* __restrict__ points, const int32_t* __restrict__ trinkets, MultiLod ml,
                                     int64_t n, int num_lods, int half_round, int sum, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c[3] = {coords[i * 3], coords[i * 3 + 1], coords[i * 3 + 2]};
    const int out_row = sum ? C : num_lods * C;
    float total[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) total[ch] = 0.0f;
    for (int l = 0; l < num_lods; ++l) {
        const int64_t p = chain[i * chain_stride + l];
        float acc[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) acc[ch] = 0.0f;
        if (p >= 0) {
            float w[8];
            trilinear_coeffs(c, points + p * 3, ml.level[l], w);
            const int32_t* tr = trinkets + p * 8;
            const T* feats = reinterpret_cast<const T*>(ml.feats[l]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const T* row = feats + (int64_t)tr[j] * C;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    float fv = Cvt<T>::to_f(row[ch]);
                    if (half_round) fv = __half2float(__float2half_rn(fv));
                    acc[ch] += fv * w[j];
                }
            }
            if (half_round)
#pragma unroll
                for (int ch = 0; ch < C; ++ch) acc[ch] = __half2float(__float2half_rn(acc[ch]));
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            if (sum) total[ch] += acc[ch]; else out[i * out_row + l * C + ch] = acc[ch];
        }
    }
    if (sum)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) out[i * out_row + ch] = total[ch];
}

extern "C" int wisp_spc_trilinear_multi_fwd(const float* coords, const int64_t* chain, int64_t chain_stride,
                                            const int16_t* points, const int32_t* trinkets, const void* const* feats,
                                            int dtype, int64_t num_samples, int num_lods, const int32_t* levels,
                                            int channels, int half_round, int sum, float* out, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0 && num_lods >= 1 && num_lods <= SPC_MAX_LODS && channels >= 1 && chain_stride >= num_lods, "bad sizes");
    if (num_samples == 0) return WISP_OK;
    WISP_REQUIRE(coords && chain && points && trinkets && feats && levels && out, "null pointer");
    MultiLod ml;
    WISP_REQUIRE(fill_multi(ml, feats, nullptr, levels, num_lods) == 0, "bad level or null feature pointer");
    hipStream_t s = (hipStream_t)stream;
    if (channels <= 8 && num_samples >= 4096) {
        const dim3 grid((unsigned)ceil_div64(num_samples, 256)), block(256);
#define TRI_SMALL(T, CC) hipLaunchKernelGGL((spc_trilinear_multi_fwd_small_kernel<T, CC>), grid, block, 0, s, coords, chain, chain_stride, \
                                             points, trinkets, ml, num_samples, num_lods, half_round, sum, out)
#define TRI_SMALL_T(CC) case CC: if (dtype == WISP_F32) TRI_SMALL(float, CC); else if (dtype == WISP_F16) TRI_SMALL(__half, CC); \
                                 else TRI_SMALL(__hip_bfloat16, CC); break;
        switch (channels) { TRI_SMALL_T(1) TRI_SMALL_T(2) TRI_SMALL_T(3) TRI_SMALL_T(4) TRI_SMALL_T(5) TRI_SMALL_T(6) TRI_SMALL_T(7) TRI_SMALL_T(8) }
#undef TRI_SMALL_T
#undef TRI_SMALL
        WISP_CHECK_LAUNCH();
        return WISP_OK;
    }
    const dim3 grid(interp_grid(num_samples, channels)), block(256);
#define TRI_MULTI(T) hipLaunchKernelGGL((spc_trilinear_multi_fwd_kernel<T>), grid, block, 0, s, coords, chain, chain_stride, points, \
                                         trinkets, ml, num_samples, num_lods, channels, half_round, sum, out)
    if (dtype == WISP_F32) TRI_MULTI(float); else if (dtype == WISP_F16) TRI_MULTI(__half); else TRI_MULTI(__hip_bfloat16);
#undef TRI_MULTI
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- VQAD codebook lookup
// Fused CodebookOctreeGrid._index_features + trilinear blend (wisp/models/grids/codebook_grid.py:103-172).  The
// reference materialises logits[S, 8, 2^bw], a softmax, a one-hot and a [S, 8, 2^bw, F] product per level; here one
// thread owns one (voxel, sample): for each of the 8 corners it reads the 2^bw logits row, finds the argmax, and blends
// the selected dictionary rows.  Training mode reproduces the straight-through estimator
//     keys = y_hard - stopgrad(y_soft) + y_soft        (forward value: one-hot up to one rounding of (1 - p) + p)
// and its backward: d logits = softmax-Jacobian^T (dictionary . g), d dictionary[argmax] += g.
#define CB_MAX_K 256
#define CB_MAX_F 16

template <typename I>
__global__ void __launch_bounds__(128)
codebook_trilinear_fwd_kernel(const float* __restrict__ coords, const I* __restrict__ pidx, const int16_t* __restrict__ points,
                              const int32_t* __restrict__ trinkets, const float* __restrict__ logits,
                              const float* __restrict__ dictionary, int64_t n, int spv, int K, int F, int level, int training,
                              float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * spv) return;
    const int64_t p = (int64_t)pidx[i / spv];
    float acc[CB_MAX_F];
    for (int f = 0; f < F; ++f) acc[f] = 0.0f;
    if (p >= 0) {
        float w[8];
        trilinear_coeffs(coords + i * 3, points + p * 3, level, w);
        for (int j = 0; j < 8; ++j) {
            const float* row = logits + (int64_t)trinkets[p * 8 + j] * K;
            int best = 0;
            float mx = row[0];
            for (int k = 1; k < K; ++k) { const float v = row[k]; if (v > mx) { mx = v; best = k; } }   // first max wins (torch.max)
            float scale = 1.0f;
            if (training) {
                float denom = 0.0f;
                for (int k = 0; k < K; ++k) denom += expf(row[k] - mx);
                const float pb = 1.0f / denom;                    // softmax probability of the argmax
                scale = (1.0f - pb) + pb;
                best = codebook_softmax_pick(row, best, mx, pb);
            }
            const float* drow = dictionary + (int64_t)best * F;
            for (int f = 0; f < F; ++f) acc[f] += drow[f] * scale * w[j];
        }
    }
    for (int f = 0; f < F; ++f) out[i * F + f] = acc[f];
}

extern "C" int wisp_codebook_trilinear_fwd(const float* coords, const void* pidx, int pidx_is_i64, const int16_t* points,
                                           const int32_t* trinkets, const float* logits, const float* dictionary,
                                           int64_t num_voxels, int samples_per_voxel, int dict_size, int feature_dim, int level,
                                           int training, float* out, wisp_stream_t stream) {
    WISP_REQUIRE(num_voxels >= 0 && samples_per_voxel >= 1 && level >= 0 && level <= 15, "bad sizes");
    WISP_REQUIRE(dict_size >= 1 && dict_size <= CB_MAX_K && feature_dim >= 1 && feature_dim <= CB_MAX_F, "dictionary too large for the fused kernel");
    if (num_voxels == 0) return WISP_OK;
    WISP_REQUIRE(coords && pidx && points && trinkets && logits && dictionary && out, "null pointer");
    const int64_t rows = num_voxels * samples_per_voxel;
    const dim3 grid((unsigned)ceil_div64(rows, 128)), block(128);
    hipStream_t s = (hipStream_t)stream;
    if (pidx_is_i64)
        hipLaunchKernelGGL(codebook_trilinear_fwd_kernel<int64_t>, grid, block, 0, s, coords, (const int64_t*)pidx, points, trinkets,
                           logits, dictionary, num_voxels, samples_per_voxel, dict_size, feature_dim, level, training, out);
    else
        hipLaunchKernelGGL(codebook_trilinear_fwd_kernel<int32_t>, grid, block, 0, s, coords, (const int32_t*)pidx, points, trinkets,
                           logits, dictionary, num_voxels, samples_per_voxel, dict_size, feature_dim, level, training, out);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---- forward in two launches: decode every logits row ONCE, then the plain trilinear blend of the decoded rows.
// The fused kernel above repeats the argmax scan (and, in training, 2^bw expf for the straight-through scale) for all eight
// corners of every SAMPLE; with 16 samples per cell and corners shared between cells the same row is decoded ~100 times.
// decoded[row] = dictionary[argmax(row)] * scale(row) is exactly what the fused kernel multiplies by the corner weight, so
// wisp_spc_trilinear_fwd over `decoded` gives bit-identical features (same products, same order): 0.23 -> ~0.08 ms per level.
__global__ void __launch_bounds__(256)
codebook_decode_rows_kernel(const float* __restrict__ logits, const float* __restrict__ dictionary, int64_t rows, int K, int F,
                            int training, float* __restrict__ decoded) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* row = logits + r * K;
    int best = 0;
    float mx = row[0];
    for (int k = 1; k < K; ++k) { const float v = row[k]; if (v > mx) { mx = v; best = k; } }   // first max wins (torch.max)
    float scale = 1.0f;
    if (training) {
        float denom = 0.0f;
        for (int k = 0; k < K; ++k) denom += expf(row[k] - mx);
        const float pb = 1.0f / denom;                    // softmax probability of the argmax
        scale = (1.0f - pb) + pb;
        best = codebook_softmax_pick(row, best, mx, pb);
    }
    const float* drow = dictionary + (int64_t)best * F;
    for (int f = 0; f < F; ++f) decoded[r * F + f] = drow[f] * scale;
}

extern "C" int wisp_codebook_decode_rows(const float* logits, const float* dictionary, int64_t num_rows, int dict_size,
                                         int feature_dim, int training, float* decoded, wisp_stream_t stream) {
    WISP_REQUIRE(num_rows >= 0 && dict_size >= 1 && feature_dim >= 1, "bad sizes");
    if (num_rows == 0) return WISP_OK;
    WISP_REQUIRE(logits && dictionary && decoded, "null pointer");
    hipLaunchKernelGGL(codebook_decode_rows_kernel, dim3((unsigned)ceil_div64(num_rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, dictionary, num_rows, dict_size, feature_dim, training, decoded);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- triplanar feature pyramid
// TriplanarGrid.interpolate (wisp/models/grids/triplanar_grid.py:97-146) + TriplanarFeatureVolume.forward (:205-233): per
// level three F.grid_sample(bilinear, align_corners=True, padding_mode='reflection') lookups - plane x with (y, z), plane
// y with (x, z), plane z with (x, y); grid[..., 0] indexes the LAST (width) dimension - stacked as [x | y | z] features,
// then cat / sum over levels.  One launch for all levels and planes; planes keep torch's [fdim, R, R] parameter layout.
#define TRI_MAX_LODS 16
struct TriPlanes { const float* fm[TRI_MAX_LODS * 3]; float* grad[TRI_MAX_LODS * 3]; int32_t size[TRI_MAX_LODS]; };

// grid_sample's coordinate pipeline for align_corners=True + reflection padding: unnormalise, reflect into [0, size-1], clip
static __device__ __forceinline__ float tri_source_index(float g, int size) {
    float x = (g + 1.0f) * 0.5f * (float)(size - 1);
    const float span = (float)(size - 1);
    if (span <= 0.0f) return 0.0f;
    x = fabsf(x);
    const float flips = floorf(x / span);
    const float extra = x - flips * span;                 // fmod(x, span)
    x = (((int)flips) & 1) ? span - extra : extra;
    return fminf(fmaxf(x, 0.0f), span);
}

template <bool BWD>
__global__ void __launch_bounds__(256)
triplane_kernel(const float* __restrict__ coords, int64_t n, TriPlanes tp, int num_lods, int fdim, int sum,
                float* __restrict__ out, const float* __restrict__ grad_out) {
    const int row = sum ? 3 * fdim : num_lods * 3 * fdim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float c[3] = {coords[i * 3], coords[i * 3 + 1], coords[i * 3 + 2]};
        if (!BWD && sum)
            for (int k = 0; k < row; ++k) out[i * row + k] = 0.0f;
        for (int l = 0; l < num_lods; ++l) {
            const int R = tp.size[l];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float gx = c[p == 0 ? 1 : 0];       // grid[..., 0] -> width:  x-plane y, y-plane x, z-plane x
                const float gy = c[p == 2 ? 1 : 2];       // grid[..., 1] -> height: x-plane z, y-plane z, z-plane y
                const float ix = tri_source_index(gx, R), iy = tri_source_index(gy, R);
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
                const float tx = ix - fx, ty = iy - fy;
                const float w00 = (1.0f - tx) * (1.0f - ty), w01 = tx * (1.0f - ty), w10 = (1.0f - tx) * ty, w11 = tx * ty;
                const bool vx1 = x1 < R, vy1 = y1 < R;    // grid_sample drops out-of-bounds corners (their weight is 0 here)
                const int col = (sum ? 0 : l * 3 * fdim) + p * fdim;
                for (int ch = 0; ch < fdim; ++ch) {
                    const int64_t base = (int64_t)ch * R * R;
                    if (!BWD) {
                        const float* fm = tp.fm[l * 3 + p] + base;
                        float v = fm[y0 * R + x0] * w00;
                        if (vx1) v += fm[y0 * R + x1] * w01;
                        if (vy1) v += fm[y1 * R + x0] * w10;
                        if (vx1 && vy1) v += fm[y1 * R + x1] * w11;
                        if (sum) out[i * row + col + ch] += v; else out[i * row + col + ch] = v;
                    } else {
                        float* gm = tp.grad[l * 3 + p] + base;
                        const float g = grad_out[i * row + col + ch];
                        atomicAdd(gm + y0 * R + x0, g * w00);
                        if (vx1) atomicAdd(gm + y0 * R + x1, g * w01);
                        if (vy1) atomicAdd(gm + y1 * R + x0, g * w10);
                        if (vx1 && vy1) atomicAdd(gm + y1 * R + x1, g * w11);
                    }
                }
            }
        }
    }
}

static int fill_planes(TriPlanes& tp, const float* const* planes, float* const* grads, const int32_t* sizes, int num_lods) {
    for (int k = 0; k < TRI_MAX_LODS * 3; ++k) { tp.fm[k] = nullptr; tp.grad[k] = nullptr; }
    for (int l = 0; l < TRI_MAX_LODS; ++l) tp.size[l] = 1;
    for (int l = 0; l < num_lods; ++l) {
        if (sizes[l] < 1) return -1;
        tp.size[l] = sizes[l];
        for (int p = 0; p < 3; ++p) {
            if (planes) { if (!planes[l * 3 + p]) return -1; tp.fm[l * 3 + p] = planes[l * 3 + p]; }
            if (grads) { if (!grads[l * 3 + p]) return -1; tp.grad[l * 3 + p] = grads[l * 3 + p]; }
        }
    }
    return 0;
}

extern "C" int wisp_triplane_fwd(const float* coords, int64_t num_samples, const float* const* planes, const int32_t* sizes,
                                 int num_lods, int feature_dim, int sum, float* out, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0 && num_lods >= 1 && num_lods <= TRI_MAX_LODS && feature_dim >= 1, "bad sizes");
    if (num_samples == 0) return WISP_OK;
    WISP_REQUIRE(coords && planes && sizes && out, "null pointer");
    TriPlanes tp;
    WISP_REQUIRE(fill_planes(tp, planes, nullptr, sizes, num_lods) == 0, "bad plane size or null plane pointer");
    hipLaunchKernelGGL(triplane_kernel<false>, dim3((unsigned)min64(ceil_div64(num_samples, 256), 16384)), dim3(256), 0,
                       (hipStream_t)stream, coords, num_samples, tp, num_lods, feature_dim, sum, out, (const float*)nullptr);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_triplane_bwd(const float* coords, int64_t num_samples, const float* grad_out, const int32_t* sizes,
                                 int num_lods, int feature_dim, int sum, float* const* grad_planes, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0 && num_lods >= 1 && num_lods <= TRI_MAX_LODS && feature_dim >= 1, "bad sizes");
    if (num_samples == 0) return WISP_OK;
    WISP_REQUIRE(coords && grad_out && sizes && grad_planes, "null pointer");
    TriPlanes tp;
    WISP_REQUIRE(fill_planes(tp, nullptr, grad_planes, sizes, num_lods) == 0, "bad plane size or null gradient pointer");
    hipLaunchKernelGGL(triplane_kernel<true>, dim3((unsigned)min64(ceil_div64(num_samples, 256), 16384)), dim3(256), 0,
                       (hipStream_t)stream, coords, num_samples, tp, num_lods, feature_dim, sum, (float*)nullptr, grad_out);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- fused sphere-tracing iteration
// One marching iteration of PackedSDFTracer.trace (wisp/tracers/packed_sdf_tracer.py:118-146) INCLUDING the field query
// nef(coords=x, channels="sdf") of a NeuralSDF over an OctreeGrid (wisp/models/nefs/neural_sdf.py:120-155:
// OctreeGrid.interpolate 'sum' over the active LODs -> [position, features] -> Linear -> relu -> Linear), in one launch and
// without a host decision in between: the reference iterates ~25 masked tensor ops + a query + an MLP + `mask.any()`
// read-back per step.  16 lanes own one pack (= one ray that hit the octree):
//   * all 16 run the (pack-uniform) bookkeeping of sphere_trace_step_kernel (csrc/render.hip, same statements, same
//     order: advance, convergence tests, far plane, find_depth_bound incl. its neighbour-bound quirk, jump, new position);
//   * they walk the octree once for the new position (the voxel on every active level);
//   * lane c gathers feature channel c of the 8 corners per level: one contiguous 64-byte row per corner and group;
//   * the hidden layer is split over the lanes (hidden / 16 neurons each, weights in LDS), the output dot product is
//     reduced over the group with four DPP-class shuffles.
// Arithmetic of query and interpolation is that of spc_query_kernel / spc_trilinear_multi_fwd_kernel (so positions, cells and
// features are bit-identical to the modular path); the decoder is an fp32 fma chain in input order, which differs from
// the library GEMM of the modular path by summation order only.
#define SDF_GROUP 16
#define SDF_MAX_HIDDEN 256
struct SdfField {
    const void* feats[SPC_MAX_LODS];
    int32_t level[SPC_MAX_LODS];
    int num_lods, channels, half_round, hidden, max_level;
    const float *w1, *b1, *w2, *b2;       // [hidden, 3 + channels], [hidden], [hidden], [1]
    float scale;
};

static __device__ __forceinline__ int sdf_child_slot(int qx, int qy, int qz, int sh) {
    return (((qx >> sh) & 1) << 2) | (((qy >> sh) & 1) << 1) | ((qz >> sh) & 1);
}

template <typename T>
__global__ void __launch_bounds__(256)
sdf_trace_fused_kernel(int64_t num_packs, int first, const float* __restrict__ nug_o, const float* __restrict__ nug_d,
                       const float* __restrict__ nug_depth, const int32_t* __restrict__ nug_pidx, float dist_max, float thr_close,
                       float thr_avg, float* __restrict__ t, float* __restrict__ dist, float* __restrict__ dist_prev,
                       uint8_t* __restrict__ mask, uint8_t* __restrict__ hit, const int32_t* __restrict__ curr_in,
                       int32_t* __restrict__ curr_out, int64_t* __restrict__ curr_pidx, float* __restrict__ x,
                       const uint8_t* __restrict__ octree, const int32_t* __restrict__ exsum, const int16_t* __restrict__ points,
                       const int32_t* __restrict__ trinkets, SdfField fld, int32_t* __restrict__ any_active) {
    extern __shared__ float s_sdf[];                    // W1 [hidden][in_pad], b1, w2 | per group: 3 + channels inputs
    const int in_dim = 3 + fld.channels;
    const int in_pad = in_dim | 1;                      // odd row stride: the lanes of a group read different rows
    float* s_w1 = s_sdf;
    float* s_b1 = s_w1 + fld.hidden * in_pad;
    float* s_w2 = s_b1 + fld.hidden;
    float* s_in = s_w2 + fld.hidden;                    // [groups per block][in_dim]
    for (int e = threadIdx.x; e < fld.hidden * in_dim; e += blockDim.x) s_w1[(e / in_dim) * in_pad + e % in_dim] = fld.w1[e];
    for (int e = threadIdx.x; e < fld.hidden; e += blockDim.x) { s_b1[e] = fld.b1[e]; s_w2[e] = fld.w2[e]; }
    __syncthreads();
    const int c = threadIdx.x & (SDF_GROUP - 1);
    const int grp = threadIdx.x / SDF_GROUP;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x / SDF_GROUP) + grp;
    if (p >= num_packs) return;
    float px, py, pz;
    bool m;
    if (first) {
        m = mask[p] != 0;
        px = x[p * 3]; py = x[p * 3 + 1]; pz = x[p * 3 + 2];
    } else {
#pragma clang fp contract(off)
        const int32_t cur = curr_in[p];
        m = mask[p] != 0;
        const bool was = m;
        bool h = hit[p] != 0;
        const float dd = dist[p];
        float tt = t[p] + dd;                                                        // t += dist          (:120)
        if (m) {
            h = fabsf(dd) < thr_close;                                               // :122
            h = h || (fabsf(dd + dist_prev[p]) * 0.5f < thr_avg);                    // :123-124
            m = tt < dist_max;                                                       // :125
        }
        m = m && !h;                                                                 // :126
        const float dprev = dd;
        int32_t nxt = -1;
        if (cur > -1) {                                                              // find_depth_bound (:131)
            uint32_t i = (uint32_t)cur;
            const uint32_t stop = (p == num_packs - 1) ? (uint32_t)num_packs : (uint32_t)curr_in[p + 1];
            while (i < stop) {
                const float entry = nug_depth[2 * (int64_t)i], exit_ = nug_depth[2 * (int64_t)i + 1];
                if ((tt >= entry && tt <= exit_) || tt < entry) { nxt = (int32_t)i; break; }
                ++i;
            }
        }
        const bool keep_prev = m;                                                    // :129 runs before :132
        m = m && (nxt != -1);                                                        // :132
        const bool jumped = nxt != cur;                                              // :133
        const int32_t now = m ? nxt : cur;                                           // :134
        if (m && jumped) tt = nug_depth[2 * (int64_t)now];                           // :136
        px = nug_o[p * 3 + 0] + nug_d[p * 3 + 0] * tt;                               // :137-139 / :121
        py = nug_o[p * 3 + 1] + nug_d[p * 3 + 1] * tt;
        pz = nug_o[p * 3 + 2] + nug_d[p * 3 + 2] * tt;
        if (c == 0) {
            if (keep_prev) dist_prev[p] = dprev;
            if (m || was) { x[p * 3 + 0] = px; x[p * 3 + 1] = py; x[p * 3 + 2] = pz; }
            if (m) curr_pidx[p] = (int64_t)nug_pidx[now];
            t[p] = tt;
            mask[p] = m ? 1 : 0;
            hit[p] = h ? 1 : 0;
            curr_out[p] = now;
        }
    }
    if (!m) return;                                      // the whole group leaves together: nothing below is wave-wide
    if (c == 0 && any_active) atomicAdd(any_active, 1);
    // ---- field query at (px, py, pz): voxel on every active level (spc_query_kernel's walk)
    const int L = fld.max_level;
    const bool inside = (fabsf(px) <= 1.0f) && (fabsf(py) <= 1.0f) && (fabsf(pz) <= 1.0f);
    const float res = (float)(1 << L);
    const int top = (1 << L) - 1;
    const int qx = min((int)floorf(res * (0.5f * px + 0.5f)), top);
    const int qy = min((int)floorf(res * (0.5f * py + 0.5f)), top);
    const int qz = min((int)floorf(res * (0.5f * pz + 0.5f)), top);
    const float pos[3] = {px, py, pz};
    float feat = 0.0f;                                   // channel c, summed over the levels
    int64_t node = inside ? 0 : -1;
    int li = 0;
    for (int l = 0; l <= L && li < fld.num_lods; ++l) {
        if (l == fld.level[li]) {
            float acc = 0.0f;
            if (node >= 0) {
                float w[8];
                trilinear_coeffs(pos, points + node * 3, l, w);
                const int32_t* tr = trinkets + node * 8;
                const T* f = reinterpret_cast<const T*>(fld.feats[li]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float fv = Cvt<T>::to_f(f[(int64_t)tr[j] * fld.channels + c]);
                    if (fld.half_round) fv = __half2float(__float2half_rn(fv));
                    acc += fv * w[j];
                }
                if (fld.half_round) acc = __half2float(__float2half_rn(acc));
            }
            feat += acc;
            ++li;
        }
        if (l < L && node >= 0) {
            const int cs = sdf_child_slot(qx, qy, qz, L - 1 - l);
            const uint32_t bits = octree[node];
            node = ((bits >> cs) & 1u) ? (int64_t)exsum[node] + __popc(bits & ((2u << cs) - 1u)) : -1;
        }
    }
    // ---- decoder: in = [position, features] (neural_sdf.py: embedded position first)
    float* gin = s_in + grp * in_dim;
    if (c < 3) gin[c] = pos[c];
    gin[3 + c] = feat;
    __builtin_amdgcn_wave_barrier();                     // the group's lanes are in one wave: LDS order suffices
    float out = 0.0f;
    for (int hh = c; hh < fld.hidden; hh += SDF_GROUP) {
        const float* wr = s_w1 + hh * in_pad;
        float a = s_b1[hh];
        for (int i = 0; i < in_dim; ++i) a = __builtin_fmaf(wr[i], gin[i], a);
        out = __builtin_fmaf(s_w2[hh], fmaxf(a, 0.0f), out);
    }
#pragma unroll
    for (int d = SDF_GROUP / 2; d >= 1; d >>= 1) out += __shfl_xor(out, d, SDF_GROUP);
    if (c == 0) dist[p] = (out + fld.b2[0]) * fld.scale;
}

extern "C" int wisp_sdf_trace_step_fused(int64_t num_packs, int first, const float* nug_o, const float* nug_d, const float* nug_depth,
                                         const int32_t* nug_pidx, float dist_max, float thr_close, float thr_avg, float* t,
                                         float* dist, float* dist_prev, uint8_t* mask, uint8_t* hit, const int32_t* curr_in,
                                         int32_t* curr_out, int64_t* curr_pidx, float* x, const uint8_t* octree,
                                         const int32_t* exsum, const int16_t* points, const int32_t* trinkets,
                                         const void* const* feats, int feats_dtype, const int32_t* levels, int num_lods,
                                         int channels, int half_round, const float* w1, const float* b1, const float* w2,
                                         const float* b2, int hidden, float scale, int32_t* any_active, wisp_stream_t stream) {
    WISP_REQUIRE(num_packs >= 0 && num_lods >= 1 && num_lods <= SPC_MAX_LODS, "bad sizes");
    WISP_REQUIRE(channels == SDF_GROUP, "the fused tracer step is built for 16 feature channels (nglod_octree.yaml)");
    WISP_REQUIRE(hidden >= 1 && hidden <= SDF_MAX_HIDDEN, "hidden width out of range");
    if (num_packs == 0) return WISP_OK;
    WISP_REQUIRE(nug_o && nug_d && nug_depth && nug_pidx && t && dist && dist_prev && mask && hit && curr_in && curr_out &&
                 curr_pidx && x && octree && exsum && points && trinkets && feats && levels && w1 && b1 && w2 && b2, "null pointer");
    SdfField fld;
    for (int l = 0; l < num_lods; ++l) {
        WISP_REQUIRE(feats[l] && levels[l] >= 0 && levels[l] <= 15 && (l == 0 || levels[l] > levels[l - 1]), "bad level list");
        fld.feats[l] = feats[l]; fld.level[l] = levels[l];
    }
    fld.num_lods = num_lods; fld.channels = channels; fld.half_round = half_round; fld.hidden = hidden;
    fld.max_level = levels[num_lods - 1];
    fld.w1 = w1; fld.b1 = b1; fld.w2 = w2; fld.b2 = b2; fld.scale = scale;
    const int groups = 256 / SDF_GROUP;
    const size_t lds = ((size_t)hidden * ((3 + channels) | 1) + 2 * hidden + (size_t)groups * (3 + channels)) * 4;
    const dim3 grid((unsigned)ceil_div64(num_packs, groups)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define SDF_LAUNCH(T) hipLaunchKernelGGL((sdf_trace_fused_kernel<T>), grid, block, lds, s, num_packs, first, nug_o, nug_d, nug_depth,   \
                                         nug_pidx, dist_max, thr_close, thr_avg, t, dist, dist_prev, mask, hit, curr_in, curr_out,     \
                                         curr_pidx, x, octree, exsum, points, trinkets, fld, any_active)
    if (feats_dtype == WISP_F32) SDF_LAUNCH(float); else if (feats_dtype == WISP_F16) SDF_LAUNCH(__half); else SDF_LAUNCH(__hip_bfloat16);
#undef SDF_LAUNCH
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- blend of gathered corners
// wisp._C.ops.grid_interpolate_cuda / _backward_cuda (wisp/csrc/ops/grid_interpolate_cuda.cu:17-129): trilinear blend of
// eight already-gathered corner rows, feats [N, 8, F] with LOCAL coordinates [N, 3] in [0, 1]; corner k = dx<<2 | dy<<1 | dz.
// Peripheral in the reference (only tests/core/test_grid_interpolation.py calls it); kept so that the whole `ops` surface
// binds.  One thread per (sample, feature): consecutive lanes read consecutive features of a corner row.  The backward
// writes every (sample, corner, feature) exactly once, so it stores instead of the reference's atomicAdd.
template <typename T>
__global__ void __launch_bounds__(256)
grid_interpolate_kernel(const float* __restrict__ coords, const T* __restrict__ feats, int64_t n, int F, T* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * F) return;
    const int64_t i = e / F;
    const int j = (int)(e - i * F);
    const float x = coords[i * 3], y = coords[i * 3 + 1], z = coords[i * 3 + 2];
    const float gx = 1.0f - x, gy = 1.0f - y, gz = 1.0f - z;
    const float c[8] = {gx * gy * gz, gx * gy * z, gx * y * gz, gx * y * z, x * gy * gz, x * gy * z, x * y * gz, x * y * z};
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += Cvt<T>::to_f(feats[(i * 8 + k) * F + j]) * c[k];
    out[e] = Cvt<T>::from_f(acc);
}

template <typename T>
__global__ void __launch_bounds__(256)
grid_interpolate_bwd_kernel(const float* __restrict__ coords, const T* __restrict__ grad_out, int64_t n, int F,
                            T* __restrict__ grad_feats) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * F) return;
    const int64_t i = e / F;
    const int j = (int)(e - i * F);
    const float x = coords[i * 3], y = coords[i * 3 + 1], z = coords[i * 3 + 2];
    const float gx = 1.0f - x, gy = 1.0f - y, gz = 1.0f - z;
    const float c[8] = {gx * gy * gz, gx * gy * z, gx * y * gz, gx * y * z, x * gy * gz, x * gy * z, x * y * gz, x * y * z};
    const float g = Cvt<T>::to_f(grad_out[e]);
#pragma unroll
    for (int k = 0; k < 8; ++k) grad_feats[(i * 8 + k) * F + j] = Cvt<T>::from_f(g * c[k]);
}

extern "C" int wisp_grid_interpolate_fwd(const float* coords, const void* feats, int dtype, int64_t num_coords, int feature_dim,
                                         void* out, wisp_stream_t stream) {
    WISP_REQUIRE(num_coords >= 0 && feature_dim >= 1, "bad sizes");
    WISP_REQUIRE(dtype == WISP_F32 || dtype == WISP_F16 || dtype == WISP_BF16, "bad dtype");
    if (num_coords == 0) return WISP_OK;
    WISP_REQUIRE(coords && feats && out, "null pointer");
    const dim3 grid((unsigned)ceil_div64(num_coords * feature_dim, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == WISP_F32) hipLaunchKernelGGL(grid_interpolate_kernel<float>, grid, block, 0, s, coords, (const float*)feats, num_coords, feature_dim, (float*)out);
    else if (dtype == WISP_F16) hipLaunchKernelGGL(grid_interpolate_kernel<__half>, grid, block, 0, s, coords, (const __half*)feats, num_coords, feature_dim, (__half*)out);
    else hipLaunchKernelGGL(grid_interpolate_kernel<__hip_bfloat16>, grid, block, 0, s, coords, (const __hip_bfloat16*)feats, num_coords, feature_dim, (__hip_bfloat16*)out);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_grid_interpolate_bwd(const float* coords, const void* grad_out, int dtype, int64_t num_coords, int feature_dim,
                                         void* grad_feats, wisp_stream_t stream) {
    WISP_REQUIRE(num_coords >= 0 && feature_dim >= 1, "bad sizes");
    WISP_REQUIRE(dtype == WISP_F32 || dtype == WISP_F16 || dtype == WISP_BF16, "bad dtype");
    if (num_coords == 0) return WISP_OK;
    WISP_REQUIRE(coords && grad_out && grad_feats, "null pointer");
    const dim3 grid((unsigned)ceil_div64(num_coords * feature_dim, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == WISP_F32) hipLaunchKernelGGL(grid_interpolate_bwd_kernel<float>, grid, block, 0, s, coords, (const float*)grad_out, num_coords, feature_dim, (float*)grad_feats);
    else if (dtype == WISP_F16) hipLaunchKernelGGL(grid_interpolate_bwd_kernel<__half>, grid, block, 0, s, coords, (const __half*)grad_out, num_coords, feature_dim, (__half*)grad_feats);
    else hipLaunchKernelGGL(grid_interpolate_bwd_kernel<__hip_bfloat16>, grid, block, 0, s, coords, (const __hip_bfloat16*)grad_out, num_coords, feature_dim, (__hip_bfloat16*)grad_feats);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- small one-hidden-layer decoder
// out = W2 relu(W1 x + b1) + b2 with one output - the NeuralSDF decoder (wisp/models/nefs/neural_sdf.py:102-118,
// nglod_octree.yaml: 19 -> 128 -> 1), forward and backward, for the training / query path (the sphere tracer has the same
// arithmetic inlined in its fused step).  16 lanes own a sample; lane c computes hidden units c, c + 16, ... from the
// weights in LDS.  Backward: g_h = dout w2_h [a_h > 0] per owned unit; d x = sum_h g_h W1[h][:] is summed per lane over
// its units and then over the group with shuffles; dW1 / db1 / dw2 / db2 accumulate in an LDS copy per workgroup (LDS
// float atomics are slow, but the whole problem is a few thousand samples per step) that is added to the global
// gradients once at the end.  fp32 throughout, fma chains in input order.
#define DEC_MAX_IN 32
template <bool BWD>
__global__ void __launch_bounds__(256)
small_decoder_kernel(const float* __restrict__ x, int64_t n, int in_dim, int hidden, const float* __restrict__ w1,
                     const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                     float* __restrict__ out, const float* __restrict__ grad_out, float* __restrict__ grad_x,
                     float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2) {
    extern __shared__ float s_dec[];
    const int in_pad = in_dim | 1;
    const int groups = blockDim.x / SDF_GROUP;
    float* s_w1 = s_dec;                                 // [hidden][in_pad]
    float* s_b1 = s_w1 + hidden * in_pad;
    float* s_w2 = s_b1 + hidden;
    float* s_in = s_w2 + hidden;                         // [groups][in_dim]
    float* s_g1 = s_in + groups * in_dim;                // backward only: dW1 [hidden][in_pad], db1 [hidden], dw2 [hidden], db2 [1]
    float* s_gb1 = s_g1 + hidden * in_pad;
    float* s_gw2 = s_gb1 + hidden;
    float* s_gb2 = s_gw2 + hidden;
    for (int e = threadIdx.x; e < hidden * in_dim; e += blockDim.x) s_w1[(e / in_dim) * in_pad + e % in_dim] = w1[e];
    for (int e = threadIdx.x; e < hidden; e += blockDim.x) { s_b1[e] = b1[e]; s_w2[e] = w2[e]; }
    if (BWD)
        for (int e = threadIdx.x; e < hidden * in_pad + 2 * hidden + 1; e += blockDim.x) s_g1[e] = 0.0f;
    __syncthreads();
    const int c = threadIdx.x & (SDF_GROUP - 1), grp = threadIdx.x / SDF_GROUP;
    float* gin = s_in + grp * in_dim;
    const int64_t rounds = (n + groups - 1) / groups;
    for (int64_t rd = blockIdx.x; rd < rounds; rd += gridDim.x) {
        const int64_t i = rd * groups + grp;
        const bool live = i < n;
        __builtin_amdgcn_wave_barrier();
        for (int k = c; k < in_dim; k += SDF_GROUP) gin[k] = live ? x[i * in_dim + k] : 0.0f;
        __builtin_amdgcn_wave_barrier();
        const float go = (BWD && live) ? grad_out[i] : 0.0f;
        float o = 0.0f;
        float dx[DEC_MAX_IN];
#pragma unroll
        for (int k = 0; k < DEC_MAX_IN; ++k) dx[k] = 0.0f;
        for (int hh = c; hh < hidden; hh += SDF_GROUP) {
            const float* wr = s_w1 + hh * in_pad;
            float a = s_b1[hh];
            for (int k = 0; k < in_dim; ++k) a = __builtin_fmaf(wr[k], gin[k], a);
            const float r = fmaxf(a, 0.0f);
            o = __builtin_fmaf(s_w2[hh], r, o);
            if (BWD) {
                if (go != 0.0f && r > 0.0f) {
                    const float g = go * s_w2[hh];
                    atomicAdd(&s_gw2[hh], go * r);
                    atomicAdd(&s_gb1[hh], g);
#pragma unroll
                    for (int k = 0; k < DEC_MAX_IN; ++k)
                        if (k < in_dim) { atomicAdd(&s_g1[hh * in_pad + k], g * gin[k]); dx[k] = __builtin_fmaf(g, wr[k], dx[k]); }
                }
            }
        }
        if (!BWD) {
#pragma unroll
            for (int d = SDF_GROUP / 2; d >= 1; d >>= 1) o += __shfl_xor(o, d, SDF_GROUP);
            if (live && c == 0) out[i] = o + b2[0];
        } else {
#pragma unroll
            for (int k = 0; k < DEC_MAX_IN; ++k) {
                if (k < in_dim) {                            // in_dim is uniform: no divergence around the shuffles
                    float v = dx[k];
#pragma unroll
                    for (int d = SDF_GROUP / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, SDF_GROUP);
                    if (live && c == (k & (SDF_GROUP - 1))) grad_x[i * in_dim + k] = v;
                }
            }
            if (live && c == 0 && go != 0.0f) atomicAdd(s_gb2, go);
        }
    }
    if (BWD) {
        __syncthreads();
        for (int e = threadIdx.x; e < hidden * in_dim; e += blockDim.x) {
            const float v = s_g1[(e / in_dim) * in_pad + e % in_dim];
            if (v != 0.0f) atomicAdd(gw1 + e, v);
        }
        for (int e = threadIdx.x; e < hidden; e += blockDim.x) {
            if (s_gb1[e] != 0.0f) atomicAdd(gb1 + e, s_gb1[e]);
            if (s_gw2[e] != 0.0f) atomicAdd(gw2 + e, s_gw2[e]);
        }
        if (threadIdx.x == 0 && s_gb2[0] != 0.0f) atomicAdd(gb2, s_gb2[0]);
    }
}

static size_t small_decoder_lds(int in_dim, int hidden, bool bwd) {
    const size_t in_pad = in_dim | 1, groups = 256 / SDF_GROUP;
    size_t f = (size_t)hidden * in_pad + 2 * hidden + groups * in_dim;
    if (bwd) f += (size_t)hidden * in_pad + 2 * hidden + 1;
    return f * 4;
}

// Dynamic LDS beyond the 64 KB default has to be allowed per function AND per device; ask only for what a launch needs
// (the NeuralSDF decoder needs ~12 KB) and remember the largest grant of every device.
template <bool BWD>
static hipError_t small_decoder_allow_lds(size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    static size_t granted[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (bytes <= granted[dev]) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(small_decoder_kernel<BWD>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) granted[dev] = bytes;
    return e;
}

extern "C" int wisp_small_decoder_fwd(const float* x, int64_t n, int in_dim, int hidden, const float* w1, const float* b1,
                                      const float* w2, const float* b2, float* out, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0 && in_dim >= 1 && in_dim <= DEC_MAX_IN && hidden >= 1 && hidden <= SDF_MAX_HIDDEN, "bad sizes (in_dim <= 32, hidden <= 256)");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(x && w1 && b1 && w2 && b2 && out, "null pointer");
    const int64_t rounds = ceil_div64(n, 256 / SDF_GROUP);
    const size_t lds = small_decoder_lds(in_dim, hidden, false);
    if (const hipError_t attr = small_decoder_allow_lds<false>(lds)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(attr));
    hipLaunchKernelGGL(small_decoder_kernel<false>, dim3((unsigned)min64(rounds, 4096)), dim3(256), lds,
                       (hipStream_t)stream, x, n, in_dim, hidden, w1, b1, w2, b2, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_small_decoder_bwd(const float* x, int64_t n, int in_dim, int hidden, const float* w1, const float* b1,
                                      const float* w2, const float* b2, const float* grad_out, float* grad_x, float* grad_w1,
                                      float* grad_b1, float* grad_w2, float* grad_b2, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0 && in_dim >= 1 && in_dim <= DEC_MAX_IN && hidden >= 1 && hidden <= SDF_MAX_HIDDEN, "bad sizes (in_dim <= 32, hidden <= 256)");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(x && w1 && b1 && w2 && b2 && grad_out && grad_x && grad_w1 && grad_b1 && grad_w2 && grad_b2, "null pointer");
    const size_t lds = small_decoder_lds(in_dim, hidden, true);
    WISP_REQUIRE(lds <= 150 * 1024, "decoder too large for the LDS gradient copy");
    if (const hipError_t attr = small_decoder_allow_lds<true>(lds)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(attr));
    const int64_t rounds = ceil_div64(n, 256 / SDF_GROUP);
    // few, long-running workgroups: every one ends with hidden x in_dim global atomics
    hipLaunchKernelGGL(small_decoder_kernel<true>, dim3((unsigned)min64(rounds, 512)), dim3(256), lds, (hipStream_t)stream, x, n,
                       in_dim, hidden, w1, b1, w2, b2, nullptr, grad_out, grad_x, grad_w1, grad_b1, grad_w2, grad_b2);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}
