// Raymarch sample generation for gfx950: the three OctreeAS modes.
//
// Replaces OctreeAS._raymarch_ray / _raymarch_voxel / _raymarch_uniform (wisp/accelstructs/octree_as.py:188-374),
// sample_from_depth_intervals / expand_pack_boundary (wisp/ops/spc/sampling.py:35-71) and
// uniform_sample_cuda_kernel (wisp/csrc/ops/uniform_sample_cuda.cu:18-59).
//
// 'ray' mode in the reference materialises ~50 bytes per CANDIDATE sample (depth, samples, pidx, mask, deltas,
// ridx, nonzero) for R*N candidates of which a few percent survive.  Here: one wave per ray, lanes = 64
// consecutive candidate steps; the occupancy test is a single bit lookup in a Morton-ordered bitfield of the
// occupancy level (64 neighbouring steps touch a handful of cache lines); __ballot gives the hit mask and
// popcount the per-ray count; after an exclusive scan over rays the emit kernel expands the mask with
// prefix-popcount ranks and writes only the surviving samples, fully coalesced.
//
// Float expressions that decide integer outputs are evaluated with contraction OFF so that they match
// oracle/raymarch.py bit for bit (every fp32 op rounded separately).
#include "wisp_common.h"

// torch.linspace(0,1,N)[s] as evaluated by the reference's device kernel (see oracle/raymarch.py::linspace01)
static __device__ __forceinline__ float linspace01(int s, int n, float step) {
#pragma clang fp contract(off)
    if (n == 1) return 0.0f;
    if (s < n / 2) return step * (float)s;
    return 1.0f - step * (float)(n - 1 - s);
}

// octree_as.py:272-277: depth = (lin + u/N) * (far-near) + near, each op rounded to fp32
static __device__ __forceinline__ float ray_depth(int s, int n, float step, float u, float fn, float range, float near) {
#pragma clang fp contract(off)
    float d = linspace01(s, n, step) + __fdiv_rn(u, fn);
    d = d * range;
    d = d + near;
    return d;
}

// addcmul(o, d, t) = o + fl(d*t)
static __device__ __forceinline__ float axpy_unfused(float o, float d, float t) {
#pragma clang fp contract(off)
    const float p = d * t;
    return o + p;
}

static __device__ __forceinline__ bool occupied(const uint32_t* __restrict__ occ_bits, const uint8_t* __restrict__ octree,
                                                const int32_t* __restrict__ exsum, float x, float y, float z, int level) {
    const bool inside = (fabsf(x) <= 1.0f) && (fabsf(y) <= 1.0f) && (fabsf(z) <= 1.0f);
    if (!inside) return false;
    const float res = (float)(1 << level);
    const int top = (1 << level) - 1;
    const int qx = min((int)floorf(res * (0.5f * x + 0.5f)), top);
    const int qy = min((int)floorf(res * (0.5f * y + 0.5f)), top);
    const int qz = min((int)floorf(res * (0.5f * z + 0.5f)), top);
    if (occ_bits) {
        const uint32_t m = wisp_cell_bit((uint32_t)qx, (uint32_t)qy, (uint32_t)qz, level);
        return (occ_bits[m >> 5] >> (m & 31u)) & 1u;
    }
    int32_t node = 0;
    for (int l = 0; l < level; ++l) {
        const int sh = level - 1 - l;
        const int c = (((qx >> sh) & 1) << 2) | (((qy >> sh) & 1) << 1) | ((qz >> sh) & 1);
        const uint32_t bits = octree[node];
        if (!((bits >> c) & 1u)) return false;
        node = exsum[node] + __popc(bits & ((2u << c) - 1u));
    }
    return true;
}

// Coarse pre-test of one 64-candidate chunk: can the ray segment [ta, tb] touch an occupied cell of the coarse level?
// The segment's axis-aligned box (inflated by 1e-4, far above the 1e-6 rounding of o + d*t) is mapped to coarse cells with
// the SAME float expression the exact test uses one octree level per factor of two finer (floor(2^L * fl(0.5 x + 0.5)):
// scaling by a power of two is exact, so coarse index == fine index >> (L - Lc)) - every cell a candidate of the chunk can
// fall into is visited, hence skipping is conservative and the kernel's outputs do not change.
static __device__ __forceinline__ bool coarse_segment_occupied(const uint32_t* s_coarse, int lc, const float (&o)[3],
                                                               const float (&d)[3], float ta, float tb) {
    const float resc = (float)(1 << lc);
    const int top = (1 << lc) - 1;
    int i0[3], i1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pa = o[a] + d[a] * ta, pb = o[a] + d[a] * tb;
        const float mn = fminf(pa, pb) - 1e-4f, mx = fmaxf(pa, pb) + 1e-4f;
        i0[a] = min(max((int)floorf(resc * (0.5f * mn + 0.5f)), 0), top);
        i1[a] = min(max((int)floorf(resc * (0.5f * mx + 0.5f)), 0), top);
        if (i1[a] - i0[a] > 2) return true;               // chunk much longer than a coarse cell: no pre-test
    }
    for (int z = i0[2]; z <= i1[2]; ++z)
        for (int y = i0[1]; y <= i1[1]; ++y)
            for (int x = i0[0]; x <= i1[0]; ++x) {
                const uint32_t m = wisp_cell_bit((uint32_t)x, (uint32_t)y, (uint32_t)z, lc);
                if ((s_coarse[m >> 5] >> (m & 31u)) & 1u) return true;
            }
    return false;
}

#define RM_COARSE_MAX_LEVEL 5          // 32^3 bits = 4 KiB of LDS
#define RAY_WAVES 1                    // rays (waves) per workgroup of the count / emit kernels: rays differ a lot in work, and a
                                       // workgroup keeps its slots until its slowest ray is done (4 -> 1: -3 us on count, -1.5 on emit)

__global__ void __launch_bounds__(256)
raymarch_ray_count_kernel(const uint32_t* __restrict__ occ_bits, const uint8_t* __restrict__ octree,
                          const int32_t* __restrict__ exsum, const float* __restrict__ origins,
                          const float* __restrict__ dirs, int64_t num_rays, float near, float range, int n, int level,
                          const float* __restrict__ jitter, uint64_t seed, const uint32_t* __restrict__ coarse_bits,
                          int coarse_level, uint32_t* __restrict__ hitmask, int32_t* __restrict__ counts) {
    __shared__ uint32_t s_coarse[(1 << (3 * RM_COARSE_MAX_LEVEL)) / 32];
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // the ray is fetched together with the coarse occupancy words, in front of the staging barrier (behind it, a workgroup - one
    // ray - paid a second memory round trip before its first useful instruction)
    const int64_t rr = r < num_rays ? r : num_rays - 1;
    const float o3[3] = {origins[rr * 3], origins[rr * 3 + 1], origins[rr * 3 + 2]};
    const float d3[3] = {dirs[rr * 3], dirs[rr * 3 + 1], dirs[rr * 3 + 2]};
    if (coarse_bits) {
        const int cw = max(1, (1 << (3 * coarse_level)) >> 5);
        for (int e = threadIdx.x; e < cw; e += blockDim.x) s_coarse[e] = coarse_bits[e];
        __syncthreads();
    }
    if (r >= num_rays) return;
    const float ox = o3[0], oy = o3[1], oz = o3[2], dx = d3[0], dy = d3[1], dz = d3[2];
    const float step = n > 1 ? __fdiv_rn(1.0f, (float)(n - 1)) : 0.0f;
    const float fn = (float)n;
    const int words = (n + 31) >> 5;
    const uint32_t key = wisp_stream_key(seed, (uint64_t)r);            // wave-uniform
    // Conservative depth interval in which this ray can be inside the (slightly inflated) cube: 64-candidate chunks that lie
    // entirely outside it cannot contain a hit and are skipped without evaluating jitter / position / occupancy.  The
    // inflation (1e-4 in space) is far above the rounding of o + d*t (|x| < 8 -> 1e-6), so no candidate the exact test
    // would accept is ever skipped; the per-candidate test itself is unchanged.
    float t_in = -INFINITY, t_out = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (fabsf(d3[a]) > 1e-12f) {
            const float ta = (-1.0001f - o3[a]) / d3[a], tb = (1.0001f - o3[a]) / d3[a];
            t_in = fmaxf(t_in, fminf(ta, tb));
            t_out = fminf(t_out, fmaxf(ta, tb));
        } else if (fabsf(o3[a]) > 1.0001f) {
            t_out = -INFINITY;                       // parallel to the slab and outside it: never inside
        }
    }
    const float pad = 1e-3f * fabsf(range) + 1e-6f;
    const int nchunks = (n + 63) >> 6;
    int cnt = 0;
    // Chunks are classified 64 at a time, one per lane: outside the cube interval -> skipped; inside, but the coarse
    // occupancy level (LDS) shows nothing along the segment -> skipped; the survivors are evaluated by the whole wave.
    for (int cg = 0; cg < nchunks; cg += 64) {
        const int ci = cg + lane;
        bool act = ci < nchunks;
        if (act && range >= 0.0f) {
            // depths of candidates base .. base+63 lie in [near + range*base/n, near + range*((base+64)/(n-1) + 1/n)]
            const int base = ci << 6;
            const float c_lo = near + range * ((float)base / fn) - pad;
            const float c_hi = near + range * ((float)(base + 64) / (float)(n > 1 ? n - 1 : 1) + 1.0f / fn) + pad;
            act = !(c_hi < t_in || c_lo > t_out);
            if (act && coarse_bits)
                act = coarse_segment_occupied(s_coarse, coarse_level, o3, d3, fmaxf(c_lo, t_in), fminf(c_hi, t_out));
        }
        if (ci < nchunks && !act) {
            const int w = ci << 1;
            hitmask[r * words + w] = 0u;
            if (w + 1 < words) hitmask[r * words + w + 1] = 0u;
        }
        unsigned long long todo = __ballot(act);
        while (todo) {
            const int c = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int base = (cg + c) << 6;
            const int s = base + lane;
            bool hit = false;
            if (s < n) {
                const float u = jitter ? jitter[r * n + s] : wisp_uniform01_keyed(key, (uint32_t)s);
                const float t = ray_depth(s, n, step, u, fn, range, near);
                hit = occupied(occ_bits, octree, exsum, axpy_unfused(ox, dx, t), axpy_unfused(oy, dy, t),
                               axpy_unfused(oz, dz, t), level);
            }
            const unsigned long long m = __ballot(hit);
            cnt += __popcll(m);
            if (lane == 0) {
                const int w = base >> 5;
                hitmask[r * words + w] = (uint32_t)m;
                if (w + 1 < words) hitmask[r * words + w + 1] = (uint32_t)(m >> 32);
            }
        }
    }
    if (lane == 0) counts[r] = cnt;
}

__global__ void __launch_bounds__(256)
raymarch_ray_emit_kernel(const float* __restrict__ origins, const float* __restrict__ dirs, int64_t num_rays,
                         float near, float range, int n, const float* __restrict__ jitter, uint64_t seed,
                         const uint32_t* __restrict__ hitmask, const int64_t* __restrict__ offsets,
                         int64_t* __restrict__ ridx, float* __restrict__ samples, float* __restrict__ depth_samples,
                         float* __restrict__ deltas, uint8_t* __restrict__ boundary, float* __restrict__ sample_dirs) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= num_rays) return;
    // offsets, the ray and the first 64 mask words in ONE memory round trip (the ray and the words used to wait for the offsets:
    // rays without samples leave below, their 24 + 256 bytes are the price)
    const int64_t begin = offsets[r], end = offsets[r + 1];
    const float ox = origins[r * 3], oy = origins[r * 3 + 1], oz = origins[r * 3 + 2];
    const float dx = dirs[r * 3], dy = dirs[r * 3 + 1], dz = dirs[r * 3 + 2];
    const int words0 = (n + 31) >> 5;
    const uint32_t first_words = lane < words0 ? hitmask[r * words0 + lane] : 0u;
    if (end == begin) return;
    const float step = n > 1 ? __fdiv_rn(1.0f, (float)(n - 1)) : 0.0f;
    const float fn = (float)n;
    const int words = (n + 31) >> 5;
    const uint32_t key = wisp_stream_key(seed, (uint64_t)r);            // wave-uniform, same stream as the count kernel
    int64_t wr = begin;
    // The ray's mask words are fetched 64 at a time with ONE coalesced load (lane = word) and only the 64-candidate chunks
    // that hold a hit are visited; walking the words one by one put a dependent load in front of every (mostly empty) chunk.
    for (int wg = 0; wg < words; wg += 64) {
        const uint32_t mine = wg == 0 ? first_words : ((wg + lane < words) ? hitmask[r * words + wg + lane] : 0u);
        const unsigned long long nz = __ballot(mine != 0u);
        unsigned long long pairs = (nz | (nz >> 1)) & 0x5555555555555555ull;     // bit 2c: chunk c of this group has a hit
        while (pairs) {
            const int b = __builtin_ctzll(pairs);                                // even word index, wave-uniform
            pairs &= pairs - 1ull;
            const unsigned long long m = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine, b) |
                                         ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine, b + 1) << 32);
            const int base = (wg + b) << 5;
            if ((m >> lane) & 1ull) {
                const int s = base + lane;
                const int64_t o = wr + __popcll(m & ((1ull << lane) - 1ull));
                const float u = jitter ? jitter[r * n + s] : wisp_uniform01_keyed(key, (uint32_t)s);
                const float t = ray_depth(s, n, step, u, fn, range, near);
                float prev = near;                                   // depth.diff(prepend=near), octree_as.py:290-291
                if (s > 0) {
                    const float up = jitter ? jitter[r * n + s - 1] : wisp_uniform01_keyed(key, (uint32_t)(s - 1));
                    prev = ray_depth(s - 1, n, step, up, fn, range, near);
                }
                ridx[o] = r;
                samples[o * 3 + 0] = axpy_unfused(ox, dx, t);
                samples[o * 3 + 1] = axpy_unfused(oy, dy, t);
                samples[o * 3 + 2] = axpy_unfused(oz, dz, t);
                depth_samples[o] = t;
                deltas[o] = t - prev;
                boundary[o] = (o == begin) ? 1 : 0;
                if (sample_dirs) { sample_dirs[o * 3 + 0] = dx; sample_dirs[o * 3 + 1] = dy; sample_dirs[o * 3 + 2] = dz; }
            }
            wr += __popcll(m);
        }
    }
}

extern "C" int wisp_raymarch_ray_count(const uint32_t* occ_bits, const uint8_t* octree, const int32_t* exsum,
                                       const float* origins, const float* dirs, int64_t num_rays, float near,
                                       float range, int num_samples, int level, const float* jitter, uint64_t seed,
                                       const uint32_t* coarse_bits, int coarse_level, uint32_t* hitmask, int32_t* counts,
                                       wisp_stream_t stream) {
    WISP_REQUIRE(num_rays >= 0 && num_samples >= 1 && level >= 0 && level <= 15, "bad sizes");
    if (num_rays == 0) return WISP_OK;
    WISP_REQUIRE(origins && dirs && hitmask && counts, "null pointer");
    WISP_REQUIRE(occ_bits || (exsum && (octree || level == 0)), "need occ_bits or octree+exsum");
    WISP_REQUIRE(!occ_bits || level <= 10, "bitfield path supports level <= 10");
    WISP_REQUIRE(!coarse_bits || (coarse_level >= 0 && coarse_level <= RM_COARSE_MAX_LEVEL && coarse_level <= level),
                 "coarse_level out of range");
    hipLaunchKernelGGL(raymarch_ray_count_kernel, dim3((unsigned)ceil_div64(num_rays, RAY_WAVES)), dim3(64 * RAY_WAVES), 0,
                       (hipStream_t)stream, occ_bits, octree, exsum, origins, dirs, num_rays, near, range, num_samples,
                       level, jitter, seed, coarse_bits, coarse_level, hitmask, counts);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_raymarch_ray_emit(const float* origins, const float* dirs, int64_t num_rays, float near, float range,
                                      int num_samples, const float* jitter, uint64_t seed, const uint32_t* hitmask,
                                      const int64_t* offsets, int64_t* ridx, float* samples, float* depth_samples,
                                      float* deltas, uint8_t* boundary, float* sample_dirs, wisp_stream_t stream) {
    WISP_REQUIRE(num_rays >= 0 && num_samples >= 1, "bad sizes");
    if (num_rays == 0) return WISP_OK;
    WISP_REQUIRE(origins && dirs && hitmask && offsets, "null pointer");
    hipLaunchKernelGGL(raymarch_ray_emit_kernel, dim3((unsigned)ceil_div64(num_rays, RAY_WAVES)), dim3(64 * RAY_WAVES), 0,
                       (hipStream_t)stream, origins, dirs, num_rays, near, range, num_samples, jitter, seed, hitmask,
                       offsets, ridx, samples, depth_samples, deltas, boundary, sample_dirs);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- 'voxel'
// sampling.py:35-55: steps = (k + u) * (1/N); depth = entry + (exit - entry) * steps
static __device__ __forceinline__ float voxel_depth(float entry, float exit_, int k, float u, float inv_n) {
#pragma clang fp contract(off)
    float st = (float)k + u;
    st = st * inv_n;
    const float span = exit_ - entry;
    const float p = span * st;
    return entry + p;
}

__global__ void __launch_bounds__(256)
raymarch_voxel_kernel(const float* __restrict__ origins, const float* __restrict__ dirs,
                      const int32_t* __restrict__ nug_ridx, const float* __restrict__ nug_depth, int64_t m, int n,
                      float inv_n, const float* __restrict__ jitter, uint64_t seed, int64_t* __restrict__ ridx,
                      float* __restrict__ samples, float* __restrict__ depth_samples, float* __restrict__ deltas,
                      uint8_t* __restrict__ boundary) {
    const int64_t total = m * n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t g = i / n;
        const int k = (int)(i - g * n);
        const int32_t r = nug_ridx[g];
        const float entry = nug_depth[g * 2], exit_ = nug_depth[g * 2 + 1];
        const float u = jitter ? jitter[i] : wisp_uniform01(seed, (uint64_t)g, (uint64_t)k);
        const float t = voxel_depth(entry, exit_, k, u, inv_n);
        float prev = entry;                                        // diff(prepend=entry), octree_as.py:220
        if (k > 0) {
            const float up = jitter ? jitter[i - 1] : wisp_uniform01(seed, (uint64_t)g, (uint64_t)(k - 1));
            prev = voxel_depth(entry, exit_, k - 1, up, inv_n);
        }
        ridx[i] = r;
        samples[i * 3 + 0] = axpy_unfused(origins[(int64_t)r * 3 + 0], dirs[(int64_t)r * 3 + 0], t);
        samples[i * 3 + 1] = axpy_unfused(origins[(int64_t)r * 3 + 1], dirs[(int64_t)r * 3 + 1], t);
        samples[i * 3 + 2] = axpy_unfused(origins[(int64_t)r * 3 + 2], dirs[(int64_t)r * 3 + 2], t);
        depth_samples[i] = t;
        deltas[i] = t - prev;
        boundary[i] = (k == 0 && (g == 0 || nug_ridx[g - 1] != r)) ? 1 : 0;   // expand_pack_boundary(mark_first_hit)
    }
}

extern "C" int wisp_raymarch_voxel_emit(const float* origins, const float* dirs, const int32_t* nug_ridx,
                                        const float* nug_depth, int64_t num_nuggets, int num_samples,
                                        const float* jitter, uint64_t seed, int64_t* ridx, float* samples,
                                        float* depth_samples, float* deltas, uint8_t* boundary, wisp_stream_t stream) {
    WISP_REQUIRE(num_nuggets >= 0 && num_samples >= 1, "bad sizes");
    if (num_nuggets == 0) return WISP_OK;
    WISP_REQUIRE(origins && dirs && nug_ridx && nug_depth && ridx && samples && depth_samples && deltas && boundary,
                 "null pointer");
    const int64_t total = num_nuggets * num_samples;
    const float inv_n = (float)(1.0 / (double)num_samples);       // python: steps *= (1.0 / num_samples)
    hipLaunchKernelGGL(raymarch_voxel_kernel, dim3((unsigned)min64(ceil_div64(total, 256), 16384)), dim3(256), 0,
                       (hipStream_t)stream, origins, dirs, nug_ridx, nug_depth, num_nuggets, num_samples, inv_n, jitter,
                       seed, ridx, samples, depth_samples, deltas, boundary);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- 'uniform'
__global__ void __launch_bounds__(256)
raymarch_uniform_count_kernel(const float* __restrict__ nug_depth, int64_t m, float scale, int32_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    // octree_as.py:340-342: ceil(scale*entry).int(), ceil(scale*exit).int()
    const int ia = (int)ceilf(scale * nug_depth[i * 2]);
    const int ib = (int)ceilf(scale * nug_depth[i * 2 + 1]);
    counts[i] = ib - ia;
}

static __device__ __forceinline__ float lattice_depth(float inv_scale, float first, float f) {
#pragma clang fp contract(off)
    const float s = first + f;
    return inv_scale * s;
}

__global__ void __launch_bounds__(256)
raymarch_uniform_emit_kernel(const float* __restrict__ origins, const float* __restrict__ dirs,
                             const int32_t* __restrict__ nug_ridx, const float* __restrict__ nug_depth, int64_t m,
                             float scale, float inv_scale, const int64_t* __restrict__ sample_offsets,
                             const int64_t* __restrict__ ray_first, int64_t* __restrict__ ridx,
                             float* __restrict__ samples, float* __restrict__ depth_samples,
                             uint8_t* __restrict__ boundary) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int64_t base = sample_offsets[i];
    const int n = (int)(sample_offsets[i + 1] - base);
    if (n <= 0) return;
    const int32_t r = nug_ridx[i];
    const float first = ceilf(scale * nug_depth[i * 2]);                     // uniform_sample_cuda.cu:46
    // first non-empty nugget of its ray <=> no sample of this ray has been emitted before `base`
    bool bval = base == sample_offsets[ray_first[r]];
    const float ox = origins[(int64_t)r * 3], oy = origins[(int64_t)r * 3 + 1], oz = origins[(int64_t)r * 3 + 2];
    const float dx = dirs[(int64_t)r * 3], dy = dirs[(int64_t)r * 3 + 1], dz = dirs[(int64_t)r * 3 + 2];
    float f = 0.0f;
    for (int k = 0; k < n; ++k) {
        const float t = lattice_depth(inv_scale, first, f);
        f += 1.0f;
        const int64_t o = base + k;
        ridx[o] = r;
        depth_samples[o] = t;
        samples[o * 3 + 0] = axpy_unfused(ox, dx, t);
        samples[o * 3 + 1] = axpy_unfused(oy, dy, t);
        samples[o * 3 + 2] = axpy_unfused(oz, dz, t);
        boundary[o] = bval ? 1 : 0;
        bval = false;
    }
}

extern "C" int wisp_raymarch_uniform_count(const float* nug_depth, int64_t num_nuggets, float scale, int32_t* counts,
                                           wisp_stream_t stream) {
    WISP_REQUIRE(num_nuggets >= 0, "negative count");
    if (num_nuggets == 0) return WISP_OK;
    WISP_REQUIRE(nug_depth && counts, "null pointer");
    hipLaunchKernelGGL(raymarch_uniform_count_kernel, dim3((unsigned)ceil_div64(num_nuggets, 256)), dim3(256), 0,
                       (hipStream_t)stream, nug_depth, num_nuggets, scale, counts);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_raymarch_uniform_emit(const float* origins, const float* dirs, const int32_t* nug_ridx,
                                          const float* nug_depth, int64_t num_nuggets, float scale,
                                          const int64_t* sample_offsets, const int64_t* ray_first, int64_t* ridx,
                                          float* samples, float* depth_samples, uint8_t* boundary,
                                          wisp_stream_t stream) {
    WISP_REQUIRE(num_nuggets >= 0, "negative count");
    if (num_nuggets == 0) return WISP_OK;
    WISP_REQUIRE(origins && dirs && nug_ridx && nug_depth && sample_offsets && ray_first, "null pointer");
    const float inv_scale = 1.0f / scale;                                    // uniform_sample_cuda.cu:86
    hipLaunchKernelGGL(raymarch_uniform_emit_kernel, dim3((unsigned)ceil_div64(num_nuggets, 256)), dim3(256), 0,
                       (hipStream_t)stream, origins, dirs, nug_ridx, nug_depth, num_nuggets, scale, inv_scale,
                       sample_offsets, ray_first, ridx, samples, depth_samples, boundary);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---- wisp._C.ops.uniform_sample_cuda under its own signature (wisp/csrc/ops/uniform_sample.cpp:28-42, kernel
// uniform_sample_cuda.cu:18-59): nuggets already filtered to non-empty ones, `insum` = inclusive sum of their sample counts.
// The marching path above (count + scan + emit with positions) does not go through this entry; it exists so that reference
// code calling the op by name binds unchanged.
__global__ void __launch_bounds__(256)
uniform_sample_kernel(int64_t m, float scale, float inv_scale, const int32_t* __restrict__ ridx, const float* __restrict__ depth,
                      const int32_t* __restrict__ insum, int64_t* __restrict__ new_ridx, float* __restrict__ depth_samples,
                      uint8_t* __restrict__ boundary) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int base = i > 0 ? insum[i - 1] : 0;
    const int n = insum[i] - base;
    const int32_t r = ridx[i];
    const float first = ceilf(scale * depth[i * 2]);
    bool bval = i == 0 || ridx[i - 1] != r;
    float f = 0.0f;
    for (int k = 0; k < n; ++k) {
        depth_samples[base + k] = lattice_depth(inv_scale, first, f);
        f += 1.0f;
        new_ridx[base + k] = r;
        boundary[base + k] = bval ? 1 : 0;
        bval = false;
    }
}

extern "C" int wisp_uniform_sample(int scale, const int32_t* ridx, const float* depth, const int32_t* insum,
                                   int64_t num_nuggets, int64_t* new_ridx, float* depth_samples, uint8_t* boundary,
                                   wisp_stream_t stream) {
    WISP_REQUIRE(num_nuggets >= 0 && scale > 0, "bad sizes");
    if (num_nuggets == 0) return WISP_OK;
    WISP_REQUIRE(ridx && depth && insum && new_ridx && depth_samples && boundary, "null pointer");
    hipLaunchKernelGGL(uniform_sample_kernel, dim3((unsigned)ceil_div64(num_nuggets, 256)), dim3(256), 0, (hipStream_t)stream,
                       num_nuggets, (float)scale, 1.0f / (float)scale, ridx, depth, insum, new_ridx, depth_samples, boundary);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}
