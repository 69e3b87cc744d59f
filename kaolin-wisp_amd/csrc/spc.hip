// SPC octree leaves for gfx950: point query, Morton occupancy bitfield, ray/octree intersection,
// pack-boundary utilities and the prefix sums the two-phase (count -> scan -> emit) kernels need.
//
// Replaces the Kaolin-Core ops wisp calls at wisp/accelstructs/octree_as.py:162 (unbatched_query),
// :183-185 (unbatched_raytrace), :300/:228 (mark_pack_boundaries / mark_first_hit) and :351
// (inclusive_sum_cuda).  Semantics: SURVEY.md Appendix A; float operation order: oracle/spc.py.
#include "wisp_common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------- query
// child slot of quantised point q at depth l of a `level`-deep walk: xbit<<2 | ybit<<1 | zbit
static __device__ __forceinline__ int child_slot(int qx, int qy, int qz, int sh) {
    return (((qx >> sh) & 1) << 2) | (((qy >> sh) & 1) << 1) | ((qz >> sh) & 1);
}

static __device__ __forceinline__ bool quantize_inside(float x, float y, float z, int level, int& qx, int& qy, int& qz) {
    // |x| <= 1 on every axis (NaN fails), q = min(floor(2^level * fl(0.5x + 0.5)), 2^level - 1)
    const bool inside = (fabsf(x) <= 1.0f) && (fabsf(y) <= 1.0f) && (fabsf(z) <= 1.0f);
    const float res = (float)(1 << level);
    const int top = (1 << level) - 1;
    qx = min((int)floorf(res * (0.5f * x + 0.5f)), top);
    qy = min((int)floorf(res * (0.5f * y + 0.5f)), top);
    qz = min((int)floorf(res * (0.5f * z + 0.5f)), top);
    return inside;
}

__global__ void __launch_bounds__(256)
spc_query_kernel(const uint8_t* __restrict__ octree, const int32_t* __restrict__ exsum,
                 const float* __restrict__ coords, int64_t n, int level, int with_parents,
                 int64_t* __restrict__ pidx) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int qx, qy, qz;
        const bool inside = quantize_inside(coords[i * 3 + 0], coords[i * 3 + 1], coords[i * 3 + 2], level, qx, qy, qz);
        int64_t node = inside ? 0 : -1;
        int64_t* row = with_parents ? pidx + i * (level + 1) : nullptr;
        if (row) row[0] = node;
        for (int l = 0; l < level; ++l) {
            if (node >= 0) {
                const int c = child_slot(qx, qy, qz, level - 1 - l);
                const uint32_t bits = octree[node];
                node = ((bits >> c) & 1u) ? (int64_t)exsum[node] + __popc(bits & ((2u << c) - 1u)) : -1;
            }
            if (row) row[l + 1] = node;
        }
        if (!row) pidx[i] = node;
    }
}

extern "C" int wisp_spc_query(const uint8_t* octree, const int32_t* exsum, const float* coords, int64_t n, int level,
                              int with_parents, int64_t* pidx, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0 && level >= 0 && level <= 15, "bad n / level");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(exsum && coords && pidx && (octree || level == 0), "null pointer");
    const int grid = (int)min64(ceil_div64(n, 256), 8192);
    hipLaunchKernelGGL(spc_query_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, octree, exsum, coords, n, level,
                       with_parents, pidx);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// The parent chain of a query, levels first_level .. level only, optionally started from a HINT: the cell of `first_level` a
// caller already knows for a group of consecutive coordinates (the 'voxel' march generates `group` samples inside every cell its
// raytrace returned).  A hint is only trusted when the coordinate really quantises into that cell at first_level; otherwise (and
// without hints) the walk starts at the root as in spc_query_kernel, so the result is the same numbers either way:
// out[i][k] = spc_query(with_parents)[i][first_level + k].
__global__ void __launch_bounds__(256)
spc_query_chain_kernel(const uint8_t* __restrict__ octree, const int32_t* __restrict__ exsum, const int16_t* __restrict__ points,
                       const float* __restrict__ coords, int64_t n, int level, int first_level,
                       const int32_t* __restrict__ hint, int group, int64_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int width = level - first_level + 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int qx, qy, qz;
        const bool inside = quantize_inside(coords[i * 3 + 0], coords[i * 3 + 1], coords[i * 3 + 2], level, qx, qy, qz);
        int64_t node = inside ? 0 : -1;
        int l = 0;
        if (hint && inside) {
            const int32_t h = hint[i / group];
            if (h >= 0) {
                const int sh = level - first_level;
                const int16_t* pt = points + (int64_t)h * 3;
                if ((qx >> sh) == (int)pt[0] && (qy >> sh) == (int)pt[1] && (qz >> sh) == (int)pt[2]) { node = h; l = first_level; }
            }
        }
        for (; l < first_level; ++l) {
            if (node >= 0) {
                const int c = child_slot(qx, qy, qz, level - 1 - l);
                const uint32_t bits = octree[node];
                node = ((bits >> c) & 1u) ? (int64_t)exsum[node] + __popc(bits & ((2u << c) - 1u)) : -1;
            }
        }
        int64_t* row = out + i * width;
        row[0] = node;
        for (; l < level; ++l) {
            if (node >= 0) {
                const int c = child_slot(qx, qy, qz, level - 1 - l);
                const uint32_t bits = octree[node];
                node = ((bits >> c) & 1u) ? (int64_t)exsum[node] + __popc(bits & ((2u << c) - 1u)) : -1;
            }
            row[l + 1 - first_level] = node;
        }
    }
}

extern "C" int wisp_spc_query_chain(const uint8_t* octree, const int32_t* exsum, const int16_t* points, const float* coords,
                                    int64_t n, int level, int first_level, const int32_t* hint_pidx, int hint_group,
                                    int64_t* chain, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0 && level >= 0 && level <= 15 && first_level >= 0 && first_level <= level, "bad n / levels");
    WISP_REQUIRE(!hint_pidx || (hint_group >= 1 && points), "a hint needs its group size and the point hierarchy");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(exsum && coords && chain && (octree || level == 0), "null pointer");
    const int grid = (int)min64(ceil_div64(n, 256), 8192);
    hipLaunchKernelGGL(spc_query_chain_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, octree, exsum, points, coords, n,
                       level, first_level, hint_pidx, hint_group > 0 ? hint_group : 1, chain);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- bitfield
__global__ void __launch_bounds__(256)
spc_bitfield_kernel(const int16_t* __restrict__ pts, int64_t n, int level, uint32_t* __restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = wisp_cell_bit((uint32_t)pts[i * 3], (uint32_t)pts[i * 3 + 1], (uint32_t)pts[i * 3 + 2], level);
    atomicOr(bits + (m >> 5), 1u << (m & 31u));
}

extern "C" int wisp_spc_build_bitfield(const int16_t* level_points, int64_t n_points, int level, uint32_t* bits,
                                       wisp_stream_t stream) {
    WISP_REQUIRE(level >= 0 && level <= 10 && bits && n_points >= 0, "level must be in [0,10]");
    const int64_t cells = (int64_t)1 << (3 * level);
    const int64_t words = (cells + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(bits, 0, words * 4, s) != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, __func__, "memset");
    if (n_points == 0) return WISP_OK;
    WISP_REQUIRE(level_points, "null points");
    hipLaunchKernelGGL(spc_bitfield_kernel, dim3((unsigned)ceil_div64(n_points, 256)), dim3(256), 0, s, level_points,
                       n_points, level, bits);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- raytrace
static bool env_flag_spc(const char* name, bool dflt) {
    const char* e = getenv(name);
    if (!e || !e[0]) return dflt;
    return e[0] != '0';
}
#define RT_MAX_LEVEL 15
#define RT_BLOCK 128

struct Slab { bool hit; float entry, exit; float cx, cy, cz; };

// float32 slab test, operation order fixed by oracle/spc.py::slab_test (see comment there)
static __device__ __forceinline__ Slab slab_test(float ox, float oy, float oz, float ix, float iy, float iz,
                                                 int px, int py, int pz, int level) {
    Slab s;
    const float r = 1.0f / (float)(1 << level);
    s.cx = r * (2.0f * (float)px + 1.0f) - 1.0f;     // exact in fp32 (power-of-two scaling)
    s.cy = r * (2.0f * (float)py + 1.0f) - 1.0f;
    s.cz = r * (2.0f * (float)pz + 1.0f) - 1.0f;
    const float t0x = ((s.cx - r) - ox) * ix, t1x = ((s.cx + r) - ox) * ix;
    const float t0y = ((s.cy - r) - oy) * iy, t1y = ((s.cy + r) - oy) * iy;
    const float t0z = ((s.cz - r) - oz) * iz, t1z = ((s.cz + r) - oz) * iz;
    const float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    const float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    s.entry = fmaxf(tmin, 0.0f);
    s.exit = tmax;
    s.hit = tmax > s.entry;
    return s;
}

// Depth-first traversal with children visited in (i XOR code) order, code from the ray ORIGIN's octant
// relative to the node centre.  EMIT = false: count only.
template <bool EMIT>
__global__ void __launch_bounds__(RT_BLOCK)
spc_raytrace_kernel(const uint8_t* __restrict__ octree, const int16_t* __restrict__ points,
                    const int32_t* __restrict__ exsum, const float* __restrict__ origins,
                    const float* __restrict__ dirs, int64_t num_rays, int level, const int64_t* __restrict__ offsets,
                    int with_exit, int32_t* __restrict__ counts, int32_t* __restrict__ out_ridx,
                    int32_t* __restrict__ out_pidx, float* __restrict__ out_depth) {
    __shared__ int32_t s_node[RT_MAX_LEVEL + 1][RT_BLOCK];
    __shared__ uint32_t s_state[RT_MAX_LEVEL + 1][RT_BLOCK];   // iter (4b) | code (3b) << 4 | bits (8b) << 8
    const int t = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * RT_BLOCK + t;
    if (r >= num_rays) return;
    const float ox = origins[r * 3], oy = origins[r * 3 + 1], oz = origins[r * 3 + 2];
    const float ix = __fdiv_rn(1.0f, dirs[r * 3]), iy = __fdiv_rn(1.0f, dirs[r * 3 + 1]), iz = __fdiv_rn(1.0f, dirs[r * 3 + 2]);
    int64_t wr = EMIT ? offsets[r] : 0;
    int32_t cnt = 0;

    Slab s = slab_test(ox, oy, oz, ix, iy, iz, 0, 0, 0, 0);
    if (s.hit) {
        if (level == 0) {
            if (EMIT) {
                out_ridx[wr] = (int32_t)r; out_pidx[wr] = 0;
                if (with_exit) { out_depth[wr * 2] = s.entry; out_depth[wr * 2 + 1] = s.exit; } else out_depth[wr] = s.entry;
            }
            cnt = 1;
        } else {
            int l = 0;
            uint32_t code = ((ox > s.cx) ? 4u : 0u) | ((oy > s.cy) ? 2u : 0u) | ((oz > s.cz) ? 1u : 0u);
            s_node[0][t] = 0;
            s_state[0][t] = 0u | (code << 4) | ((uint32_t)octree[0] << 8);
            while (l >= 0) {
                const uint32_t st = s_state[l][t];
                uint32_t it = st & 15u;
                if (it >= 8u) { --l; continue; }
                const uint32_t cd = (st >> 4) & 7u, bits = st >> 8;
                const uint32_t j = it ^ cd;
                s_state[l][t] = st + 1u;                       // advance iterator
                if (!((bits >> j) & 1u)) continue;
                const int32_t node = s_node[l][t];
                const int32_t child = exsum[node] + __popc(bits & ((2u << j) - 1u));
                const int px = points[(int64_t)child * 3], py = points[(int64_t)child * 3 + 1], pz = points[(int64_t)child * 3 + 2];
                const Slab c = slab_test(ox, oy, oz, ix, iy, iz, px, py, pz, l + 1);
                if (!c.hit) continue;
                if (l + 1 == level) {
                    if (EMIT) {
                        out_ridx[wr] = (int32_t)r; out_pidx[wr] = child;
                        if (with_exit) { out_depth[wr * 2] = c.entry; out_depth[wr * 2 + 1] = c.exit; } else out_depth[wr] = c.entry;
                        ++wr;
                    }
                    ++cnt;
                } else {
                    ++l;
                    const uint32_t ccode = ((ox > c.cx) ? 4u : 0u) | ((oy > c.cy) ? 2u : 0u) | ((oz > c.cz) ? 1u : 0u);
                    s_node[l][t] = child;
                    s_state[l][t] = 0u | (ccode << 4) | ((uint32_t)octree[child] << 8);
                }
            }
        }
    }
    if (!EMIT) counts[r] = cnt;
}

// ---- 8 lanes per ray.  The thread-per-ray walk above is a chain of dependent loads (octree byte -> exsum -> point) per CHILD
// with ~1.5 waves per SIMD at a 100 K-ray batch: pure latency (0.31 ms per pass for 102 K rays of the SynV8 level-7 tree).
// Here a group of 8 lanes owns a ray and tests the 8 children of the current node at once - lane i takes child
// i XOR code, so lane order IS the reference's visiting order; siblings are contiguous in the point hierarchy, i.e. one
// coalesced read - and the DFS advances one NODE per step.  All hits of a leaf parent are emitted in one step (rank =
// popcount of the lower lanes of the group's hit mask).  Results are the same nuggets in the same order as the walk above.
// MODE 0 = count, and park up to `cap` nuggets per ray in `cache` ([R, cap, 3]: pidx bits, entry, exit) so that the emit
// phase is a copy instead of a second traversal; MODE 1 = traverse and write (rays whose nuggets did not fit the cache).
#define RT8_RAYS 32                       // rays per workgroup (256 threads)
template <int MODE>
__global__ void __launch_bounds__(RT8_RAYS * 8)
spc_raytrace8_kernel(const uint8_t* __restrict__ octree, const int16_t* __restrict__ points,
                     const int32_t* __restrict__ exsum, const float* __restrict__ origins,
                     const float* __restrict__ dirs, int64_t num_rays, int level, const int64_t* __restrict__ offsets,
                     int with_exit, int32_t* __restrict__ counts, float* __restrict__ cache, int cap,
                     int32_t* __restrict__ out_ridx, int32_t* __restrict__ out_pidx, float* __restrict__ out_depth) {
    __shared__ int32_t s_base[RT_MAX_LEVEL][RT8_RAYS];
    __shared__ uint32_t s_st[RT_MAX_LEVEL][RT8_RAYS];        // pending hit mask (8b) | code (3b) << 8 | child bits (8b) << 16
    const int g = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const int shift = (threadIdx.x & 63) & ~7;               // position of this group's 8 bits inside the wave ballot
    const int64_t r = (int64_t)blockIdx.x * RT8_RAYS + g;
    bool active = r < num_rays;
    int64_t wr = 0;
    if (MODE == 1 && active) {
        wr = offsets[r];
        if (cache && offsets[r + 1] - wr <= (int64_t)cap) active = false;      // served from the cache by the copy kernel
    }
    float ox = 0.f, oy = 0.f, oz = 0.f, ix = 0.f, iy = 0.f, iz = 0.f;
    if (active) {
        ox = origins[r * 3]; oy = origins[r * 3 + 1]; oz = origins[r * 3 + 2];
        ix = __fdiv_rn(1.0f, dirs[r * 3]); iy = __fdiv_rn(1.0f, dirs[r * 3 + 1]); iz = __fdiv_rn(1.0f, dirs[r * 3 + 2]);
    }
    int32_t cnt = 0;
    int l = 0;
    uint32_t bits = 0, code = 0;
    int32_t base = 0;
    if (active) {
        const Slab s = slab_test(ox, oy, oz, ix, iy, iz, 0, 0, 0, 0);
        if (!s.hit) active = false;
        else if (level == 0) {
            if (sub == 0) {
                if (MODE == 1) {
                    out_ridx[wr] = (int32_t)r; out_pidx[wr] = 0;
                    if (with_exit) { out_depth[wr * 2] = s.entry; out_depth[wr * 2 + 1] = s.exit; } else out_depth[wr] = s.entry;
                } else if (cache && cap > 0) {
                    float* c = cache + (int64_t)r * cap * 3;
                    c[0] = __int_as_float(0); c[1] = s.entry; c[2] = s.exit;
                }
            }
            cnt = 1;
            active = false;
        } else {
            code = ((ox > s.cx) ? 4u : 0u) | ((oy > s.cy) ? 2u : 0u) | ((oz > s.cz) ? 1u : 0u);
            bits = octree[0];
            base = exsum[0];
        }
    }
    while (__any(active)) {
        // ---- test the 8 children of the current node (depth l -> l + 1)
        bool hit = false;
        Slab c;
        int32_t child = 0;
        if (active) {
            const uint32_t j = (uint32_t)sub ^ code;
            if ((bits >> j) & 1u) {
                child = base + __popc(bits & ((2u << j) - 1u));
                const int16_t* pt = points + (int64_t)child * 3;
                c = slab_test(ox, oy, oz, ix, iy, iz, pt[0], pt[1], pt[2], l + 1);
                hit = c.hit;
            }
        }
        uint32_t mask = (uint32_t)(__ballot(hit) >> shift) & 0xffu;
        if (active) {
            if (l + 1 == level) {                              // leaf parent: every hit is a nugget, in lane order
                if (hit) {
                    const int k = __popc(mask & ((1u << sub) - 1u));
                    if (MODE == 1) {
                        const int64_t w = wr + k;
                        out_ridx[w] = (int32_t)r; out_pidx[w] = child;
                        if (with_exit) { out_depth[w * 2] = c.entry; out_depth[w * 2 + 1] = c.exit; } else out_depth[w] = c.entry;
                    } else if (cache && cnt + k < cap) {
                        float* cc = cache + ((int64_t)r * cap + cnt + k) * 3;
                        cc[0] = __int_as_float(child); cc[1] = c.entry; cc[2] = c.exit;
                    }
                }
                const int m = __popc(mask);
                cnt += m; wr += m;
                mask = 0;
            }
            // ---- next node: first pending child here, else back up
            while (mask == 0u && l > 0) {
                --l;
                const uint32_t st = s_st[l][g];
                mask = st & 0xffu; code = (st >> 8) & 7u; bits = st >> 16;
                base = s_base[l][g];
            }
            if (mask == 0u) active = false;
            else {
                const uint32_t i0 = (uint32_t)__builtin_ctz(mask);
                mask &= mask - 1u;
                s_st[l][g] = mask | (code << 8) | (bits << 16);      // all 8 lanes write the same value
                s_base[l][g] = base;
                const uint32_t j0 = i0 ^ code;
                const int32_t node = base + __popc(bits & ((2u << j0) - 1u));
                ++l;
                bits = octree[node];
                base = exsum[node];
                const int16_t* pt = points + (int64_t)node * 3;
                const float rr = 1.0f / (float)(1 << l);
                const float cx = rr * (2.0f * (float)pt[0] + 1.0f) - 1.0f, cy = rr * (2.0f * (float)pt[1] + 1.0f) - 1.0f,
                            cz = rr * (2.0f * (float)pt[2] + 1.0f) - 1.0f;
                code = ((ox > cx) ? 4u : 0u) | ((oy > cy) ? 2u : 0u) | ((oz > cz) ? 1u : 0u);
            }
        }
    }
    if (MODE == 0 && r < num_rays && sub == 0) counts[r] = cnt;
}

// emit phase for the rays whose nuggets were parked by the count phase: 8 lanes copy one ray's run
__global__ void __launch_bounds__(RT8_RAYS * 8)
spc_raytrace_copy_kernel(const float* __restrict__ cache, int cap, int64_t num_rays, const int64_t* __restrict__ offsets,
                         int with_exit, int32_t* __restrict__ out_ridx, int32_t* __restrict__ out_pidx,
                         float* __restrict__ out_depth) {
    const int64_t r = (int64_t)blockIdx.x * RT8_RAYS + (threadIdx.x >> 3);
    if (r >= num_rays) return;
    const int64_t w0 = offsets[r];
    const int n = (int)(offsets[r + 1] - w0);
    if (n > cap) return;                                      // re-traversed by spc_raytrace8_kernel<1>
    const float* c = cache + (int64_t)r * cap * 3;
    for (int k = threadIdx.x & 7; k < n; k += 8) {
        const int64_t w = w0 + k;
        out_ridx[w] = (int32_t)r;
        out_pidx[w] = __float_as_int(c[k * 3]);
        if (with_exit) { out_depth[w * 2] = c[k * 3 + 1]; out_depth[w * 2 + 1] = c[k * 3 + 2]; } else out_depth[w] = c[k * 3 + 1];
    }
}

static bool rt_use_groups() { static const bool v = env_flag_spc("WISP_RAYTRACE_GROUPS", true); return v; }

extern "C" int wisp_spc_raytrace_count(const uint8_t* octree, const int16_t* points, const int32_t* exsum,
                                       const float* origins, const float* dirs, int64_t num_rays, int level,
                                       int32_t* counts, float* cache, int cache_cap, wisp_stream_t stream) {
    WISP_REQUIRE(num_rays >= 0 && level >= 0 && level <= RT_MAX_LEVEL && cache_cap >= 0, "bad num_rays / level / cache_cap");
    if (num_rays == 0) return WISP_OK;
    WISP_REQUIRE(points && exsum && origins && dirs && counts && (octree || level == 0), "null pointer");
    if (rt_use_groups())
        hipLaunchKernelGGL(spc_raytrace8_kernel<0>, dim3((unsigned)ceil_div64(num_rays, RT8_RAYS)), dim3(RT8_RAYS * 8), 0,
                           (hipStream_t)stream, octree, points, exsum, origins, dirs, num_rays, level, nullptr, 0, counts,
                           cache, cache ? cache_cap : 0, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL(spc_raytrace_kernel<false>, dim3((unsigned)ceil_div64(num_rays, RT_BLOCK)), dim3(RT_BLOCK), 0,
                           (hipStream_t)stream, octree, points, exsum, origins, dirs, num_rays, level, nullptr, 0, counts,
                           nullptr, nullptr, nullptr);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_spc_raytrace_emit(const uint8_t* octree, const int16_t* points, const int32_t* exsum,
                                      const float* origins, const float* dirs, int64_t num_rays, int level,
                                      const int64_t* offsets, int with_exit, const float* cache, int cache_cap,
                                      int32_t* ridx, int32_t* pidx, float* depth, wisp_stream_t stream) {
    WISP_REQUIRE(num_rays >= 0 && level >= 0 && level <= RT_MAX_LEVEL && cache_cap >= 0, "bad num_rays / level / cache_cap");
    if (num_rays == 0) return WISP_OK;
    WISP_REQUIRE(points && exsum && origins && dirs && offsets && ridx && pidx && depth && (octree || level == 0),
                 "null pointer");
    if (rt_use_groups()) {
        const bool cached = cache != nullptr && cache_cap > 0;
        if (cached)
            hipLaunchKernelGGL(spc_raytrace_copy_kernel, dim3((unsigned)ceil_div64(num_rays, RT8_RAYS)), dim3(RT8_RAYS * 8), 0,
                               (hipStream_t)stream, cache, cache_cap, num_rays, offsets, with_exit, ridx, pidx, depth);
        hipLaunchKernelGGL(spc_raytrace8_kernel<1>, dim3((unsigned)ceil_div64(num_rays, RT8_RAYS)), dim3(RT8_RAYS * 8), 0,
                           (hipStream_t)stream, octree, points, exsum, origins, dirs, num_rays, level, offsets, with_exit,
                           nullptr, const_cast<float*>(cached ? cache : nullptr), cached ? cache_cap : 0, ridx, pidx, depth);
    } else {
        hipLaunchKernelGGL(spc_raytrace_kernel<true>, dim3((unsigned)ceil_div64(num_rays, RT_BLOCK)), dim3(RT_BLOCK), 0,
                           (hipStream_t)stream, octree, points, exsum, origins, dirs, num_rays, level, offsets, with_exit,
                           nullptr, ridx, pidx, depth);
    }
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- pack boundaries
template <typename I>
__global__ void __launch_bounds__(256)
mark_boundaries_kernel(const I* __restrict__ ids, int64_t n, uint8_t* __restrict__ b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    b[i] = (i == 0 || ids[i] != ids[i - 1]) ? 1 : 0;
}

extern "C" int wisp_mark_pack_boundaries_i64(const int64_t* ids, int64_t n, uint8_t* boundary, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(ids && boundary, "null pointer");
    hipLaunchKernelGGL(mark_boundaries_kernel<int64_t>, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, ids, n, boundary);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_mark_pack_boundaries_i32(const int32_t* ids, int64_t n, uint8_t* boundary, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(ids && boundary, "null pointer");
    hipLaunchKernelGGL(mark_boundaries_kernel<int32_t>, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, ids, n, boundary);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- scans
// Three-phase scan: (1) per-tile sums, (2) one workgroup scans the tile sums, (3) per-tile rescan with the
// tile's base.  Tile = 256 threads x 8 items.  Wave-level scan by DPP-friendly __shfl_up over 64 lanes.
#define SC_THREADS 256
#define SC_ITEMS 8
#define SC_TILE (SC_THREADS * SC_ITEMS)

static __device__ __forceinline__ int64_t wave_incl_scan(int64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread workgroup; returns exclusive prefix, *total = sum
static __device__ __forceinline__ int64_t block_excl_scan(int64_t v, int64_t* total) {
    __shared__ int64_t s_w[SC_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t inc = wave_incl_scan(v, lane);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int64_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SC_THREADS / 64; ++k) {
        const int64_t x = s_w[k];
        if (k < w) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <typename IN>
__global__ void __launch_bounds__(SC_THREADS)
scan_tile_sums_kernel(const IN* __restrict__ in, int64_t n, int64_t* __restrict__ tile_sums) {
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k)
        if (base + k < n) s += (int64_t)in[base + k];
    int64_t tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// in-place exclusive scan of tile_sums by ONE workgroup; writes the grand total to *total_out (may be null)
__global__ void __launch_bounds__(SC_THREADS)
scan_spine_kernel(int64_t* __restrict__ tile_sums, int64_t nt, int64_t* __restrict__ total_out) {
    int64_t carry = 0;
    for (int64_t b = 0; b < nt; b += SC_THREADS) {
        const int64_t i = b + threadIdx.x;
        const int64_t v = i < nt ? tile_sums[i] : 0;
        int64_t tot;
        const int64_t ex = block_excl_scan(v, &tot);
        if (i < nt) tile_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename IN, typename OUT, bool INCLUSIVE>
__global__ void __launch_bounds__(SC_THREADS)
scan_apply_kernel(const IN* __restrict__ in, int64_t n, const int64_t* __restrict__ tile_base, OUT* __restrict__ out) {
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    int64_t v[SC_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        v[k] = (base + k < n) ? (int64_t)in[base + k] : 0;
        s += v[k];
    }
    int64_t tot;
    int64_t run = block_excl_scan(s, &tot) + tile_base[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        if (base + k < n) out[base + k] = (OUT)(INCLUSIVE ? run + v[k] : run);
        run += v[k];
    }
}

extern "C" int64_t wisp_scan_workspace_bytes(int64_t n) {
    const int64_t nt = n > 0 ? ceil_div64(n, SC_TILE) : 1;
    return (nt + 8) * 8;
}

// Up to 64 K values (the per-ray sample counts of one training batch): ONE launch, every workgroup independent.
// Workgroup b owns values [4096 b, 4096 b + 4096) and simply re-sums everything in front of them (coalesced 16-byte loads of
// at most 240 KiB that sit in L2) instead of waiting for anybody: no spine kernel, no look-back flags, and the longest
// workgroup reads what ONE workgroup of a serial scan would have read anyway.  (History: a one-workgroup version took 89 us
// for 50 K values with strided per-thread runs and still 37 us with coalesced tiles - one CU cannot move the data faster;
// a launch costs ~5 us of timeline on this stack, so the three-kernel tile scan below is kept for long inputs only.)
#define SC1_THREADS 1024
#define SC1_TILE (SC1_THREADS * 4)
#define SC1_MAX_TILES 16
#define SC1_MAX (SC1_TILE * SC1_MAX_TILES)

// block-wide exclusive scan of one int64 per thread (SC1_THREADS threads); *total = sum over the workgroup
static __device__ __forceinline__ int64_t sc1_block_excl_scan(int64_t v, int64_t* total) {
    __shared__ int64_t s_w[SC1_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t inc = wave_incl_scan(v, lane);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int64_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SC1_THREADS / 64; ++k) {
        const int64_t x = s_w[k];
        if (k < w) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(SC1_THREADS)
scan_small_kernel(const int32_t* __restrict__ in, int n, int64_t* __restrict__ out) {
    // everything in front of this tile: at most SC1_MAX_TILES - 1 tiles, one 16-byte load per thread and tile, issued
    // eight at a time before any is consumed (a rolled loop waits for every load in turn)
    const int4* __restrict__ in4 = reinterpret_cast<const int4*>(in);
    int64_t part = 0;
    if (blockIdx.x > 0) {                        // (tile 0 is complete then: it is the safe address for masked-off loads)
#pragma unroll
        for (int k0 = 0; k0 < SC1_MAX_TILES; k0 += 8) {                      // eight loads in flight, then eight more
            int4 q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {                                    // no branch per load: a branch ends the
                const unsigned tile = (k0 + k < (int)blockIdx.x) ? (unsigned)(k0 + k) : 0u;   // basic block and its wait
                q[k] = in4[tile * SC1_THREADS + threadIdx.x];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < (int)blockIdx.x) part += ((int64_t)q[k].x + q[k].y) + ((int64_t)q[k].z + q[k].w);
        }
    }
    const int i = blockIdx.x * SC1_TILE + 4 * (int)threadIdx.x;
    int32_t v[4] = {0, 0, 0, 0};
    if (i + 3 < n) {
        const int4 t = *reinterpret_cast<const int4*>(in + i);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (i + e < n) v[e] = in[i + e];
    }
    int64_t base;
    sc1_block_excl_scan(part, &base);                                     // base = sum of everything before this tile
    int64_t tot;
    const int64_t e0 = base + sc1_block_excl_scan((int64_t)v[0] + v[1] + v[2] + v[3], &tot);
    const int64_t e1 = e0 + v[0], e2 = e1 + v[1], e3 = e2 + v[2];
    if (i + 3 < n) {
        longlong2* dst = reinterpret_cast<longlong2*>(out + i);
        dst[0] = make_longlong2(e0, e1);
        dst[1] = make_longlong2(e2, e3);
    } else {
        const int64_t ex[4] = {e0, e1, e2, e3};
#pragma unroll
        for (int e = 0; e < 4; ++e) if (i + e < n) out[i + e] = ex[e];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = base + tot;
}

extern "C" int wisp_exclusive_scan_i32(const int32_t* counts, int64_t n, int64_t* offsets, void* workspace,
                                       wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0 && offsets, "bad args");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (hipMemsetAsync(offsets, 0, 8, s) != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, __func__, "memset");
        return WISP_OK;
    }
    WISP_REQUIRE(counts && workspace, "null pointer");
    if (n <= SC1_MAX) {
        hipLaunchKernelGGL(scan_small_kernel, dim3((unsigned)ceil_div64(n, SC1_TILE)), dim3(SC1_THREADS), 0, s, counts, (int)n, offsets);
        WISP_CHECK_LAUNCH();
        return WISP_OK;
    }
    int64_t* tiles = (int64_t*)workspace;
    const int64_t nt = ceil_div64(n, SC_TILE);
    hipLaunchKernelGGL(scan_tile_sums_kernel<int32_t>, dim3((unsigned)nt), dim3(SC_THREADS), 0, s, counts, n, tiles);
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(SC_THREADS), 0, s, tiles, nt, offsets + n);
    hipLaunchKernelGGL((scan_apply_kernel<int32_t, int64_t, false>), dim3((unsigned)nt), dim3(SC_THREADS), 0, s, counts,
                       n, tiles, offsets);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_inclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, void* workspace,
                                       wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(in && out && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    int64_t* tiles = (int64_t*)workspace;
    const int64_t nt = ceil_div64(n, SC_TILE);
    hipLaunchKernelGGL(scan_tile_sums_kernel<int32_t>, dim3((unsigned)nt), dim3(SC_THREADS), 0, s, in, n, tiles);
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(SC_THREADS), 0, s, tiles, nt, (int64_t*)nullptr);
    hipLaunchKernelGGL((scan_apply_kernel<int32_t, int32_t, true>), dim3((unsigned)nt), dim3(SC_THREADS), 0, s, in, n,
                       tiles, out);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- boundary -> pack starts
__global__ void __launch_bounds__(SC_THREADS)
boundary_tile_counts_kernel(const uint8_t* __restrict__ b, int64_t n, int32_t* __restrict__ counts) {
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k)
        if (base + k < n) s += b[base + k] ? 1 : 0;
    int64_t tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) counts[blockIdx.x] = (int32_t)tot;
}

__global__ void __launch_bounds__(SC_THREADS)
boundary_pack_starts_kernel(const uint8_t* __restrict__ b, int64_t n, const int64_t* __restrict__ tile_offsets,
                            int64_t* __restrict__ starts) {
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    uint8_t f[SC_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        f[k] = (base + k < n) ? b[base + k] : 0;
        s += f[k] ? 1 : 0;
    }
    int64_t tot;
    int64_t run = block_excl_scan(s, &tot) + tile_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k)
        if (f[k]) starts[run++] = base + k;
}

extern "C" int wisp_boundary_tile_counts(const uint8_t* boundary, int64_t n, int32_t* counts, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(boundary && counts, "null pointer");
    hipLaunchKernelGGL(boundary_tile_counts_kernel, dim3((unsigned)ceil_div64(n, SC_TILE)), dim3(SC_THREADS), 0,
                       (hipStream_t)stream, boundary, n, counts);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_boundary_pack_starts(const uint8_t* boundary, int64_t n, const int64_t* tile_offsets,
                                         int64_t* starts, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(boundary && tile_offsets && starts, "null pointer");
    hipLaunchKernelGGL(boundary_pack_starts_kernel, dim3((unsigned)ceil_div64(n, SC_TILE)), dim3(SC_THREADS), 0,
                       (hipStream_t)stream, boundary, n, tile_offsets, starts);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- SPC build on the device
// Replaces the Kaolin-Core build the reference runs at construction and inside every prune
// (unbatched_points_to_octree + scan_octrees + generate_points: wisp/ops/spc/conversions.py:29-40,72-88,
// wisp/models/nefs/nerf.py:205-206): sort + unique + level-by-level compaction there.  Here the cells of the finest level
// are a DENSE mask in Morton order (1 byte per cell: 2 MiB at level 7, 16 MiB at level 8 - nothing on this GPU) and the
// hierarchy falls out of it without any sort:
//   * the byte of a level-l node IS eight consecutive entries of the level-(l+1) occupancy (child c = x<<2 | y<<1 | z are
//     the low three Morton bits), so every level's node bytes come from the level below with one 8-byte load per node;
//   * the concatenation [dense node bytes of levels 0..L-1 | leaf mask] in that order is the point hierarchy in BFS /
//     Morton order with holes; ONE stream compaction of its non-zero entries (the boundary -> pack-starts kernels above)
//     yields every point of every level in its final position;
//   * a point's coordinates are the Morton decode of its dense index.
static __device__ __forceinline__ uint32_t morton_spread3(uint32_t v) {       // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
static __device__ __forceinline__ uint32_t morton_compact3(uint32_t v) {      // inverse of morton_spread3
    v &= 0x09249249u;
    v = (v | (v >> 2)) & 0x030C30C3u;
    v = (v | (v >> 4)) & 0x0300F00Fu;
    v = (v | (v >> 8)) & 0x030000FFu;
    v = (v | (v >> 16)) & 0x000003FFu;
    return v;
}

__global__ void __launch_bounds__(256)
spc_mask_from_points_kernel(const int16_t* __restrict__ points, int64_t n, int level, uint8_t* __restrict__ leaf_mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t res = 1u << level;
    const uint32_t x = (uint32_t)points[i * 3], y = (uint32_t)points[i * 3 + 1], z = (uint32_t)points[i * 3 + 2];
    if (x >= res || y >= res || z >= res) return;                              // also drops negative coordinates
    leaf_mask[(morton_spread3(x) << 2) | (morton_spread3(y) << 1) | morton_spread3(z)] = 1;   // duplicates write the same value
}

// out[i] = sum_c (in[8 i + c] != 0) << c  : node bytes of one level from the occupancy (or node bytes) of the level below
__global__ void __launch_bounds__(256)
spc_parent_bytes_kernel(const uint8_t* __restrict__ in, int64_t n_parents, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents) return;
    // a level starts at element (8^l - 1) / 7 of the dense array = 1 mod 8: the eight bytes are NOT 8-byte aligned
    uint64_t w;
    __builtin_memcpy(&w, in + i * 8, 8);
    uint32_t b = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) b |= (((w >> (8 * c)) & 0xffull) != 0 ? 1u : 0u) << c;
    out[i] = (uint8_t)b;
}

__global__ void __launch_bounds__(256)
spc_points_from_index_kernel(const int64_t* __restrict__ index, int64_t n, int level, int16_t* __restrict__ points) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t d = index[i];                                    // position in [level 0 | level 1 | ... | level L]
    int l = 0;
    while (l < level && d >= ((int64_t)1 << (3 * l))) { d -= (int64_t)1 << (3 * l); ++l; }
    const uint32_t m = (uint32_t)d;                          // Morton code inside level l
    points[i * 3 + 0] = (int16_t)morton_compact3(m >> 2);
    points[i * 3 + 1] = (int16_t)morton_compact3(m >> 1);
    points[i * 3 + 2] = (int16_t)morton_compact3(m);
}

extern "C" int wisp_spc_mask_from_points(const int16_t* points, int64_t n, int level, uint8_t* leaf_mask, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0 && level >= 1 && level <= 10, "bad sizes (level 1..10)");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(points && leaf_mask, "null pointer");
    hipLaunchKernelGGL(spc_mask_from_points_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       points, n, level, leaf_mask);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_spc_dense_bytes(uint8_t* dense, int level, wisp_stream_t stream) {
    // dense: u8 [(8^(level+1) - 1) / 7]: levels 0..level concatenated; the caller filled the last 8^level entries (leaf
    // mask, Morton order); the node bytes of levels level-1 .. 0 are written in front of it.
    WISP_REQUIRE(dense && level >= 1 && level <= 10, "bad arguments (level 1..10)");
    int64_t off[12];
    off[0] = 0;
    for (int l = 0; l <= level; ++l) off[l + 1] = off[l] + ((int64_t)1 << (3 * l));
    for (int l = level - 1; l >= 0; --l) {
        const int64_t n = (int64_t)1 << (3 * l);
        hipLaunchKernelGGL(spc_parent_bytes_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                           dense + off[l + 1], n, dense + off[l]);
    }
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_spc_points_from_index(const int64_t* index, int64_t n, int level, int16_t* points, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0 && level >= 0 && level <= 10, "bad sizes");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(index && points, "null pointer");
    hipLaunchKernelGGL(spc_points_from_index_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       index, n, level, points);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}
