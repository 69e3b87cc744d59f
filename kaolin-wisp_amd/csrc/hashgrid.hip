// Multi-resolution hash / dense grid interpolation for gfx950 (MI355X).
//
// Replaces wisp/csrc/ops/hashgrid_interpolate_cuda.cu:19-339 + hash_utils.cuh:17-112 (reference design:
// one launch per level, one thread per sample, uncoalesced 4-byte stores).  Design here:
//   * ONE launch for all levels.  A workgroup owns a tile of 64 consecutive samples; wave w of the
//     workgroup owns level w (w, w+16, ... when there are more than 16 levels).  All per-level state
//     (resolution, dense-vs-hash, table base) is therefore wave-uniform: it lives in SGPRs and the
//     dense/hash branch never diverges.
//   * consecutive samples of a ray are consecutive lanes, so on the coarse (dense) levels a wave's 8x64
//     corner reads fall into a handful of cache lines.
//   * the [64 x L*F] output tile is staged through LDS and written back as whole rows: every global
//     store instruction writes 256 contiguous bytes per wave (the reference scatters 4-byte pieces at a
//     64-byte stride).
//   * backward accumulates into an fp32 gradient table with hardware global_atomic_add_f32.
// Numerics contract: SURVEY.md Appendix B (double-then-float coordinate scaling, float32 clamp bound,
// uint32 hash with the reference's primes, float32 blend in corner order).
#include "wisp_common.h"
#include <stdlib.h>

#define HG_MAX_LODS 32
#define HG_TILE 64

struct HashLevels {
    int32_t res[HG_MAX_LODS];
    int32_t dense[HG_MAX_LODS];
};

template <int DIM>
struct CornerSetup {
    int32_t idx[1 << DIM];
    float coef[1 << DIM];
};

// Position / coefficient / index computation shared by forward and backward.
template <int DIM>
static __device__ __forceinline__ void corner_setup(const float* __restrict__ c, int32_t res, bool dense,
                                                    uint32_t tsize, bool tsize_pow2, CornerSetup<DIM>& cs) {
    const float hi = (float)((double)(res - 1) - 1e-5);          // hashgrid_interpolate_cuda.cu:40, clamp bound
    int32_t pos[DIM];
    float f[DIM], g[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        double xd = (double)res * ((double)c[a] * 0.5 + 0.5);     // evaluated in double, rounded once to float
        float x = (float)xd;
        x = fmaxf(0.0f, fminf(hi, x));                            // hash_utils.cuh:108-112
        float p = floorf(x);
        pos[a] = (int32_t)p;
        f[a] = x - p;
        g[a] = 1.0f - f[a];
    }
#pragma unroll
    for (int j = 0; j < (1 << DIM); ++j) {
        float w = 1.0f;
        int32_t corner[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            const int bit = (j >> (DIM - 1 - a)) & 1;
            const float t = bit ? f[a] : g[a];
            w = (a == 0) ? t : w * t;                             // left-to-right product, .cu:49-56
            corner[a] = pos[a] + bit;
        }
        cs.coef[j] = w;
        int32_t idx;
        if (dense) {                                              // hash_utils.cuh:27-32
            idx = corner[0] + corner[1] * res;
            if (DIM == 3) idx += corner[2] * res * res;
        } else {                                                  // hash_utils.cuh:34-36 (uint32 wrap-around)
            uint32_t h = (uint32_t)corner[0] * 1u ^ (uint32_t)corner[1] * 2654435761u;
            if (DIM == 3) h ^= (uint32_t)corner[2] * 805459861u;
            idx = (int32_t)(tsize_pow2 ? (h & (tsize - 1u)) : (h % tsize));
        }
        cs.idx[j] = idx;
    }
}

template <typename T, int F, int DIM>
__global__ void __launch_bounds__(1024)
hashgrid_fwd_kernel(const float* __restrict__ coords, int64_t n, const T* __restrict__ codebook,
                    const int64_t* __restrict__ first_idx, HashLevels lv, int num_lods, uint32_t tsize,
                    int tsize_pow2, int zero_from_col, T* __restrict__ feats) {
    extern __shared__ __attribute__((aligned(16))) uint32_t stage[];   // [num_lods][65][W] dwords
    constexpr int W = (F * (int)sizeof(T)) / 4;                        // payload dwords per (sample, level)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int64_t ntiles = (n + HG_TILE - 1) / HG_TILE;
    const int row_dw = num_lods * W;                                   // dwords per output row

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t i = tile * HG_TILE + lane;
        const bool live = i < n;
        float c[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) c[a] = live ? coords[i * DIM + a] : 0.0f;

        for (int l = wave; l < num_lods; l += nwaves) {
            const int32_t res = __builtin_amdgcn_readfirstlane(lv.res[l]);
            const bool dense = __builtin_amdgcn_readfirstlane(lv.dense[l]) != 0;
            float acc[F];
#pragma unroll
            for (int k = 0; k < F; ++k) acc[k] = 0.0f;
            if (live && l * F < zero_from_col) {
                const T* __restrict__ table = codebook + first_idx[l] * F;
                CornerSetup<DIM> cs;
                corner_setup<DIM>(c, res, dense, tsize, tsize_pow2 != 0, cs);
                T v[1 << DIM][F];
#pragma unroll
                for (int j = 0; j < (1 << DIM); ++j) {
                    const T* p = table + (int64_t)cs.idx[j] * F;
                    if constexpr (W == 1) {
                        *reinterpret_cast<uint32_t*>(&v[j][0]) = *reinterpret_cast<const uint32_t*>(p);
                    } else if constexpr (W == 2) {
                        *reinterpret_cast<uint2*>(&v[j][0]) = *reinterpret_cast<const uint2*>(p);
                    } else if constexpr (W == 4) {
                        *reinterpret_cast<uint4*>(&v[j][0]) = *reinterpret_cast<const uint4*>(p);
                    } else {
#pragma unroll
                        for (int k = 0; k < F; ++k) v[j][k] = p[k];
                    }
                }
#pragma unroll
                for (int j = 0; j < (1 << DIM); ++j)
#pragma unroll
                    for (int k = 0; k < F; ++k) acc[k] += Cvt<T>::to_f(v[j][k]) * cs.coef[j];
#pragma unroll
                for (int k = 0; k < F; ++k)
                    if (l * F + k >= zero_from_col) acc[k] = 0.0f;
            }
            T o[F];
#pragma unroll
            for (int k = 0; k < F; ++k) o[k] = Cvt<T>::from_f(acc[k]);
            uint32_t* dst = stage + (l * 65 + lane) * W;
#pragma unroll
            for (int w = 0; w < W; ++w) dst[w] = reinterpret_cast<const uint32_t*>(o)[w];
        }
        __syncthreads();
        // write the tile back as full rows: consecutive threads -> consecutive dwords of feats
        const int64_t rows = (n - tile * HG_TILE) < HG_TILE ? (n - tile * HG_TILE) : HG_TILE;
        const int total = (int)rows * row_dw;
        uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(feats) + tile * HG_TILE * row_dw;
        for (int gidx = threadIdx.x; gidx < total; gidx += blockDim.x) {
            const int s = gidx / row_dw;
            const int rem = gidx - s * row_dw;
            const int l = rem / W;
            const int w = rem - l * W;
            out[gidx] = stage[(l * 65 + s) * W + w];
        }
        __syncthreads();
    }
}

// Backward.  Global fp32 atomics on MI355X execute memory-side at ~1.4e10 /s no matter the scope bits, so the
// lever is the NUMBER of atomics, not their placement.  Consecutive samples of a ray are consecutive lanes and on
// every level several of them fall into the same grid cell (from ~2 per cell at res 512 to the whole wave at res
// 16 for the 2048-step march), and a cell fixes all 2^DIM corner indices.  Each wave therefore runs a segmented
// (by cell id) inclusive scan over its 64 lanes for the 2^DIM x F corner contributions and only the LAST lane of
// every run issues atomics: ~38 instead of 256 atomics per sample for the nerf_hash.yaml configuration.
template <typename T, int F, int DIM, bool MERGE>
__global__ void __launch_bounds__(1024)
hashgrid_bwd_kernel(const float* __restrict__ coords, int64_t n, const T* __restrict__ grad_feats,
                    const int64_t* __restrict__ first_idx, HashLevels lv, int num_lods, uint32_t tsize,
                    int tsize_pow2, int zero_from_col, float* __restrict__ grad_codebook) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int64_t ntiles = (n + HG_TILE - 1) / HG_TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t i = tile * HG_TILE + lane;
        const bool live = i < n;
        float c[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) c[a] = live ? coords[i * DIM + a] : 0.0f;
        for (int l = wave; l < num_lods; l += nwaves) {
            if (l * F >= zero_from_col) continue;
            const int32_t res = __builtin_amdgcn_readfirstlane(lv.res[l]);
            const bool dense = __builtin_amdgcn_readfirstlane(lv.dense[l]) != 0;
            float g[F];
#pragma unroll
            for (int k = 0; k < F; ++k) g[k] = 0.0f;
            if (live) {
                const T* gp = grad_feats + (i * num_lods + l) * F;
#pragma unroll
                for (int k = 0; k < F; ++k) g[k] = (l * F + k < zero_from_col) ? Cvt<T>::to_f(gp[k]) : 0.0f;
            }
            CornerSetup<DIM> cs;
            corner_setup<DIM>(c, res, dense, tsize, tsize_pow2 != 0, cs);
            float v[1 << DIM][F];
#pragma unroll
            for (int j = 0; j < (1 << DIM); ++j)
#pragma unroll
                for (int k = 0; k < F; ++k) v[j][k] = g[k] * cs.coef[j];
            bool issue = live;
            if (MERGE) {
                // cell id = the (dense-style) linear index of corner 0; unique per cell for res^DIM < 2^31
                int32_t key;
                {
                    const float hi = (float)((double)(res - 1) - 1e-5);
                    int32_t lin = 0, mul = 1;
#pragma unroll
                    for (int a = 0; a < DIM; ++a) {
                        float x = (float)((double)res * ((double)c[a] * 0.5 + 0.5));
                        x = fmaxf(0.0f, fminf(hi, x));
                        lin += (int32_t)floorf(x) * mul;
                        mul *= res;
                    }
                    key = live ? lin : (-2 - lane);
                }
                const int32_t prevk = __shfl_up(key, 1, 64);
                int f = (lane == 0 || key != prevk) ? 1 : 0;          // run-head flag
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int fp = __shfl_up(f, d, 64);
                    const bool take = (lane >= d) && !f;
#pragma unroll
                    for (int j = 0; j < (1 << DIM); ++j)
#pragma unroll
                        for (int k = 0; k < F; ++k) {
                            const float vp = __shfl_up(v[j][k], d, 64);
                            if (take) v[j][k] += vp;
                        }
                    if (take) f |= fp;
                }
                const int32_t nextk = __shfl_down(key, 1, 64);
                issue = live && (lane == 63 || key != nextk);         // run tail holds the run total
            }
            if (issue) {
                float* __restrict__ gt = grad_codebook + first_idx[l] * F;
#pragma unroll
                for (int j = 0; j < (1 << DIM); ++j) {
                    float* p = gt + (int64_t)cs.idx[j] * F;
#pragma unroll
                    for (int k = 0; k < F; ++k) atomicAdd(p + k, v[j][k]);   // global_atomic_add_f32
                }
            }
        }
    }
}

static int fill_levels(const int32_t* resolutions, int num_lods, int coord_dim, int64_t tsize, HashLevels& lv) {
    for (int l = 0; l < num_lods; ++l) {
        const int32_t r = resolutions[l];
        if (r < 1) return -1;
        lv.res[l] = r;
        // hash_utils.cuh:27-29 / :75-76 -- strict '<' on int32 products (wrap-around preserved)
        const int32_t ts = (int32_t)tsize;
        const int32_t r2 = (int32_t)((uint32_t)r * (uint32_t)r);
        const int32_t r3 = (int32_t)((uint32_t)r2 * (uint32_t)r);
        bool dense = (r < ts) && (r2 < ts);
        if (coord_dim == 3) dense = dense && (r3 < ts);
        lv.dense[l] = dense ? 1 : 0;
    }
    for (int l = num_lods; l < HG_MAX_LODS; ++l) { lv.res[l] = 1; lv.dense[l] = 1; }
    return 0;
}

static inline int hg_grid(int64_t n) {
    int64_t tiles = ceil_div64(n, HG_TILE);
    int64_t g = tiles < 2048 ? tiles : 2048;     // 256 CUs x 8; grid-stride beyond that
    return (int)(g < 1 ? 1 : g);
}

template <typename T, int F, int DIM>
static int launch_fwd(const float* coords, int64_t n, const void* codebook, const int64_t* first_idx,
                      const HashLevels& lv, int num_lods, uint32_t tsize, int zero_from_col, void* feats,
                      hipStream_t s) {
    constexpr int W = (F * (int)sizeof(T)) / 4;
    const int nw = num_lods < 16 ? num_lods : 16;
    const size_t lds = (size_t)num_lods * 65 * W * 4;
    const int pow2 = (tsize & (tsize - 1)) == 0;
    hipLaunchKernelGGL((hashgrid_fwd_kernel<T, F, DIM>), dim3(hg_grid(n)), dim3(64 * nw), lds, s, coords, n,
                       (const T*)codebook, first_idx, lv, num_lods, tsize, pow2, zero_from_col, (T*)feats);
    return 0;
}

static bool bwd_merge_enabled() {
    static const int v = [] { const char* e = getenv("WISP_HG_BWD_MERGE"); return (e && e[0] == '0') ? 0 : 1; }();
    return v != 0;
}

template <typename T, int F, int DIM>
static int launch_bwd(const float* coords, int64_t n, const void* grad_feats, const int64_t* first_idx,
                      const HashLevels& lv, int num_lods, uint32_t tsize, int zero_from_col, float* grad_codebook,
                      hipStream_t s) {
    const int nw = num_lods < 16 ? num_lods : 16;
    const int pow2 = (tsize & (tsize - 1)) == 0;
    // the run merge keys cells by a 32-bit linear id: fall back to plain scatter for absurd resolutions / wide features
    bool merge = bwd_merge_enabled() && (F * (1 << DIM) <= 32);
    for (int l = 0; l < num_lods; ++l) {
        double cells = 1.0;
        for (int a = 0; a < DIM; ++a) cells *= (double)lv.res[l];
        if (cells >= 2147483648.0) merge = false;
    }
    if (merge)
        hipLaunchKernelGGL((hashgrid_bwd_kernel<T, F, DIM, true>), dim3(hg_grid(n)), dim3(64 * nw), 0, s, coords, n,
                           (const T*)grad_feats, first_idx, lv, num_lods, tsize, pow2, zero_from_col, grad_codebook);
    else
        hipLaunchKernelGGL((hashgrid_bwd_kernel<T, F, DIM, false>), dim3(hg_grid(n)), dim3(64 * nw), 0, s, coords, n,
                           (const T*)grad_feats, first_idx, lv, num_lods, tsize, pow2, zero_from_col, grad_codebook);
    return 0;
}

#define HG_DISPATCH_F(T, DIM, FN, ...)                                   \
    switch (feature_dim) {                                               \
        case 2: FN<T, 2, DIM>(__VA_ARGS__); break;                       \
        case 4: FN<T, 4, DIM>(__VA_ARGS__); break;                       \
        case 8: FN<T, 8, DIM>(__VA_ARGS__); break;                       \
        case 16: FN<T, 16, DIM>(__VA_ARGS__); break;                     \
        default: return wisp_fail(WISP_ERR_UNSUPPORTED, __func__, "feature_dim must be 2, 4, 8 or 16"); \
    }

#define HG_DISPATCH(FN, ...)                                             \
    if (coord_dim == 3) {                                                \
        if (dtype == WISP_F32) { HG_DISPATCH_F(float, 3, FN, __VA_ARGS__) }              \
        else if (dtype == WISP_F16) { HG_DISPATCH_F(__half, 3, FN, __VA_ARGS__) }        \
        else { HG_DISPATCH_F(__hip_bfloat16, 3, FN, __VA_ARGS__) }                       \
    } else {                                                             \
        if (dtype == WISP_F32) { HG_DISPATCH_F(float, 2, FN, __VA_ARGS__) }              \
        else if (dtype == WISP_F16) { HG_DISPATCH_F(__half, 2, FN, __VA_ARGS__) }        \
        else { HG_DISPATCH_F(__hip_bfloat16, 2, FN, __VA_ARGS__) }                       \
    }

extern "C" int wisp_hashgrid_interpolate_fwd(const float* coords, int64_t n, int coord_dim, const void* codebook,
                                             int dtype, int feature_dim, const int64_t* first_idx,
                                             const int32_t* resolutions, int num_lods, int codebook_bitwidth,
                                             int zero_from_col, void* feats, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(coords && codebook && first_idx && resolutions && feats, "null pointer");
    WISP_REQUIRE(coord_dim == 2 || coord_dim == 3, "coord_dim must be 2 or 3");
    WISP_REQUIRE(num_lods >= 1 && num_lods <= HG_MAX_LODS, "num_lods out of range");
    WISP_REQUIRE(codebook_bitwidth >= 1 && codebook_bitwidth <= 30, "codebook_bitwidth out of range");
    WISP_REQUIRE(dtype == WISP_F32 || dtype == WISP_F16 || dtype == WISP_BF16, "bad dtype");
    HashLevels lv;
    const int64_t tsize = (int64_t)1 << codebook_bitwidth;
    WISP_REQUIRE(fill_levels(resolutions, num_lods, coord_dim, tsize, lv) == 0, "bad resolution");
    hipStream_t s = (hipStream_t)stream;
    HG_DISPATCH(launch_fwd, coords, n, codebook, first_idx, lv, num_lods, (uint32_t)tsize, zero_from_col, feats, s)
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_hashgrid_interpolate_bwd(const float* coords, int64_t n, int coord_dim, const void* grad_feats,
                                             int dtype, int feature_dim, const int64_t* first_idx,
                                             const int32_t* resolutions, int num_lods, int codebook_bitwidth,
                                             int zero_from_col, float* grad_codebook, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(coords && grad_feats && first_idx && resolutions && grad_codebook, "null pointer");
    WISP_REQUIRE(coord_dim == 2 || coord_dim == 3, "coord_dim must be 2 or 3");
    WISP_REQUIRE(num_lods >= 1 && num_lods <= HG_MAX_LODS, "num_lods out of range");
    WISP_REQUIRE(codebook_bitwidth >= 1 && codebook_bitwidth <= 30, "codebook_bitwidth out of range");
    WISP_REQUIRE(dtype == WISP_F32 || dtype == WISP_F16 || dtype == WISP_BF16, "bad dtype");
    HashLevels lv;
    const int64_t tsize = (int64_t)1 << codebook_bitwidth;
    WISP_REQUIRE(fill_levels(resolutions, num_lods, coord_dim, tsize, lv) == 0, "bad resolution");
    hipStream_t s = (hipStream_t)stream;
    HG_DISPATCH(launch_bwd, coords, n, grad_feats, first_idx, lv, num_lods, (uint32_t)tsize, zero_from_col,
                grad_codebook, s)
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}
