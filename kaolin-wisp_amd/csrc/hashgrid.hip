// Multi-resolution hash / dense grid interpolation for gfx950 (MI355X).
//
// Replaces wisp/csrc/ops/hashgrid_interpolate_cuda.cu:19-339 + hash_utils.cuh:17-112 (reference design:
// one launch per level, one thread per sample, uncoalesced 4-byte stores).  Design here:
//   * ONE launch for all levels.  A wave owns a tile of 64 consecutive samples and walks the levels; all per-level
//     state (resolution, dense-vs-hash, table base) is wave-uniform: it lives in SGPRs and the dense/hash branch
//     never diverges.  Waves are independent - no workgroup barrier anywhere.
//   * consecutive samples of a ray are consecutive lanes, so on the coarse (dense) levels a wave's 8x64
//     corner reads fall into a handful of cache lines.
//   * the [64 x L*F] output tile is staged through the wave's LDS slice and written back as one contiguous block:
//     every global store instruction writes 256 contiguous bytes per wave (the reference scatters 4-byte pieces
//     at a 64-byte stride).
//   * backward (second half of this file): run merge inside the wave, records binned per 8192 table entries, one
//     workgroup per bin accumulating in LDS - atomics only as the overflow / tiny-batch path.
// Numerics contract: SURVEY.md Appendix B (coordinate scaling rounded once from the exact value like the reference's
// double expression, float32 clamp bound, uint32 hash with the reference's primes, float32 blend in corner order).
#include "wisp_common.h"
#include <stdlib.h>

#define HG_MAX_LODS 32
#define HG_TILE 64

struct HashLevels {
    int32_t res[HG_MAX_LODS];
    int32_t dense[HG_MAX_LODS];
    // per level, computed once on the host (the same IEEE operations the kernels used to repeat per wave and level, three of
    // them in fp64): the float32 clamp bound of hashgrid_interpolate_cuda.cu:40 and res / 2
    float hi[HG_MAX_LODS];
    float hr[HG_MAX_LODS];
};

template <int DIM>
struct CornerSetup {
    int32_t idx[1 << DIM];
    float coef[1 << DIM];
    int32_t cell[DIM];          // integer coordinates of corner 0
    float frac[DIM];            // position inside the cell (read by the diagnostic wisp_hashgrid_cells only)
};

// Position / coefficient / index computation shared by forward and backward.
template <int DIM>
static __device__ __forceinline__ void corner_setup(const float* __restrict__ c, int32_t res, float hi, float hr, bool dense,
                                                    uint32_t tsize, bool tsize_pow2, CornerSetup<DIM>& cs) {
    // hi = (float)((double)(res - 1) - 1e-5): hashgrid_interpolate_cuda.cu:40, clamp bound;  hr = 0.5f * res (exact: res < 2^24)
    int32_t pos[DIM];
    float f[DIM], g[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        // reference (hash_utils.cuh:108-112): float x = res * (c * 0.5 + 0.5) evaluated in double, rounded once to float.
        // res/2 * c + res/2 is exact in double whenever |c| >= 2^-18 (<= 52 significant bits), so ONE fp32 fma - exact
        // product and sum, one rounding - returns the same float; for smaller |c| the two could only differ if the exact
        // value sat within 2^-53 relative of a float rounding midpoint.  Saves four fp64 instructions per axis and level.
        float x = __builtin_fmaf(hr, c[a], hr);
        x = fmaxf(0.0f, fminf(hi, x));
        float p = floorf(x);
        pos[a] = (int32_t)p;
        cs.cell[a] = pos[a];
        f[a] = x - p;
        cs.frac[a] = f[a];
        g[a] = 1.0f - f[a];
    }
    // Per-axis partial terms, shared by the corners: index terms for (pos, pos + 1) - (p + 1) * k == p * k + k in uint32
    // arithmetic, so one multiply per axis serves both - and the blend factors (g, f).
    uint32_t term[DIM][2];
    if (dense) {                                                  // hash_utils.cuh:27-32: x + y * res + z * res * res
        uint32_t mul = 1u;
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            term[a][0] = (uint32_t)pos[a] * mul;
            term[a][1] = term[a][0] + mul;
            mul *= (uint32_t)res;
        }
    } else {                                                      // hash_utils.cuh:34-36 (uint32 wrap-around)
        const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            term[a][0] = (uint32_t)pos[a] * primes[a];
            term[a][1] = term[a][0] + primes[a];
        }
    }
    if constexpr (DIM == 3) {
        // left-to-right products (.cu:49-56) two at a time: v_pk_mul_f32 does the (x y) pairs and then (xy z0, xy z1) =
        // coefficients j, j + 1 - six packed multiplies instead of twelve scalar ones, same roundings
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 X = {g[0], f[0]}, Z = {g[2], f[2]};
        const f32x2 xy0 = X * g[1], xy1 = X * f[1];              // (bx, by = 0), (bx, by = 1)
        const f32x2 c00 = Z * xy0[0], c01 = Z * xy1[0], c10 = Z * xy0[1], c11 = Z * xy1[1];
        cs.coef[0] = c00[0]; cs.coef[1] = c00[1]; cs.coef[2] = c01[0]; cs.coef[3] = c01[1];
        cs.coef[4] = c10[0]; cs.coef[5] = c10[1]; cs.coef[6] = c11[0]; cs.coef[7] = c11[1];
    } else {
#pragma unroll
        for (int j = 0; j < (1 << DIM); ++j) {
            float w = 1.0f;
#pragma unroll
            for (int a = 0; a < DIM; ++a) {
                const float t = ((j >> (DIM - 1 - a)) & 1) ? f[a] : g[a];
                w = (a == 0) ? t : w * t;                         // left-to-right product, .cu:49-56
            }
            cs.coef[j] = w;
        }
    }
    // the index flavour is uniform for the whole wave: branch once, not per corner
#define HG_CORNER_TERMS(OP)                                                                               \
    _Pragma("unroll") for (int j = 0; j < (1 << DIM); ++j) {                                              \
        uint32_t h = term[0][(j >> (DIM - 1)) & 1];                                                       \
        _Pragma("unroll") for (int a = 1; a < DIM; ++a) h = h OP term[a][(j >> (DIM - 1 - a)) & 1];       \
        hh[j] = h;                                                                                        \
    }
    uint32_t hh[1 << DIM];
    if (dense) {
        HG_CORNER_TERMS(+)
#pragma unroll
        for (int j = 0; j < (1 << DIM); ++j) cs.idx[j] = (int32_t)hh[j];
    } else {
        HG_CORNER_TERMS(^)
        if (tsize_pow2) {
#pragma unroll
            for (int j = 0; j < (1 << DIM); ++j) cs.idx[j] = (int32_t)(hh[j] & (tsize - 1u));
        } else {
#pragma unroll
            for (int j = 0; j < (1 << DIM); ++j) cs.idx[j] = (int32_t)(hh[j] % tsize);
        }
    }
#undef HG_CORNER_TERMS
}

// Forward.  One WAVE owns 64 consecutive samples and walks all levels for them: the level is wave-uniform (resolution,
// dense-vs-hash and table base are scalar loads, the index flavour never diverges), the waves are independent (no
// workgroup barrier: a wave of a cheap coarse level never waits for a wave of an expensive hashed one) and the tile's
// [64 x L*F] output is collected in the wave's LDS slice and written back as one contiguous block.
#define FW_WAVES 4
template <typename T, int F, int DIM>
__global__ void __launch_bounds__(FW_WAVES * 64)
hashgrid_fwd_kernel(const float* __restrict__ coords, int64_t n, const T* __restrict__ codebook,
                    const int64_t* __restrict__ first_idx, HashLevels lv, int num_lods, uint32_t tsize,
                    int tsize_pow2, int zero_from_col, int row_shift, int off32, T* __restrict__ feats) {
    extern __shared__ __attribute__((aligned(16))) uint32_t stage_all[];   // per wave: [num_lods][65][W] dwords
    constexpr int W = (F * (int)sizeof(T)) / 4;                            // payload dwords per (sample, level)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t* stage = stage_all + (size_t)wave * num_lods * 65 * W;
    const int64_t ntiles = (n + HG_TILE - 1) / HG_TILE;
    const int row_dw = num_lods * W;                                       // dwords per output row

    const int nwaves = blockDim.x >> 6;                                    // 4, fewer when the rows are wide (LDS)
    for (int64_t tile = (int64_t)blockIdx.x * nwaves + wave; tile < ntiles; tile += (int64_t)gridDim.x * nwaves) {
        const int64_t i = tile * HG_TILE + lane;
        const bool live = i < n;
        float c[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) c[a] = live ? coords[i * DIM + a] : 0.0f;

        for (int l = 0; l < num_lods; ++l) {
            const int32_t res = lv.res[l];
            const bool dense = lv.dense[l] != 0;
            float acc[F];
#pragma unroll
            for (int k = 0; k < F; ++k) acc[k] = 0.0f;
            if (live && l * F < zero_from_col) {
                const T* __restrict__ table = codebook + first_idx[l] * F;
                CornerSetup<DIM> cs;
                corner_setup<DIM>(c, res, lv.hi[l], lv.hr[l], dense, tsize, tsize_pow2 != 0, cs);
                if (dense && res >= 258) {
                    // the fp32 clamp bound rounds up to res - 1 here, so a corner can be `res` and its index
                    // can leave the level: the reference then reads the next level's rows - and past the allocation on the
                    // last level, which is the one case refused here (the read is pinned to the table's last row)
                    const int64_t last = first_idx[num_lods] - 1 - first_idx[l];
#pragma unroll
                    for (int j = 0; j < (1 << DIM); ++j)
                        if ((int64_t)(uint32_t)cs.idx[j] > last) cs.idx[j] = (int32_t)last;
                }
                T v[1 << DIM][F];
                auto fetch = [&](int j, const T* p) {
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
                };
                if (off32 && (W == 1 || W == 2 || W == 4)) {
                    // byte offsets inside the level fit 32 bits whenever the whole table is < 4 GB (checked on the host):
                    // buffer loads - scalar descriptor of the level + 32-bit vector offset, one multiply / shift per corner
                    // instead of a sign extension and a 64-bit add
                    const __amdgpu_buffer_rsrc_t rsrc =
                        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(table), (short)0, (int)0x7fffffff, 0x00020000);
                    if (W == 1 && dense && res < 258) {
                        // Dense level: the row index is x + y res + z res^2, so a corner and its +x neighbour (j and j + half:
                        // axis 0 is the top bit of j) are ADJACENT entries - one 8-byte load fetches both.  The kernel is bound
                        // by the rate at which the texture-address path takes lane requests (a table that fits L2 eight times
                        // over is only 8 % faster, scripts/exp_l2.py), and this halves them on the dense levels: 120 -> 88
                        // gathers per sample at the nerf_hash shape.
                        constexpr int half = 1 << (DIM - 1);
                        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int j = 0; j < half; ++j) {
                            const int voff = (int)((uint32_t)cs.idx[j] * (uint32_t)(F * sizeof(T)));
                            const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0);
                            *reinterpret_cast<uint32_t*>(&v[j][0]) = t[0];
                            *reinterpret_cast<uint32_t*>(&v[j + half][0]) = t[1];
                        }
                    } else if (W == 1 && !dense && tsize_pow2 != 0 && tsize >= 2u) {
                        // Hashed level, power-of-two table: the x term of the hash is x itself (prime 1) and the mask keeps its low
                        // bit, so for an EVEN cell coordinate x the corner x + 1 = x ^ 1 is the other entry of the same aligned
                        // pair: one 8-byte load per (y, z) serves both.  Lanes with an odd x fetch their +x corners afterwards
                        // (four more loads, issued for those lanes only): 6 lane requests per sample and level on average
                        // instead of 8.  Measured (scripts/bench_hashfwd.py, 2 M ray-ordered samples): 202 -> 153 us with tables
                        // of 2^14 entries per level (cache resident: the request rate was the bound), 213 -> 213 us at the
                        // 2^19 of nerf_hash.yaml - there the bound is the rate of LINE fills into the vector L1, and the
                        // +x neighbour always came out of the line its partner had just brought in.
                        constexpr int half = 1 << (DIM - 1);
                        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int j = 0; j < half; ++j) {
                            const uint32_t i0 = (uint32_t)cs.idx[j];
                            const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)((i0 & ~1u) * 4u), 0, 0);
                            const bool upper = (i0 & 1u) != 0;
                            *reinterpret_cast<uint32_t*>(&v[j][0]) = upper ? t[1] : t[0];
                            *reinterpret_cast<uint32_t*>(&v[j + half][0]) = upper ? t[0] : t[1];       // entry i0 ^ 1
                        }
                        if (cs.cell[0] & 1) {
#pragma unroll
                            for (int j = half; j < (1 << DIM); ++j)
                                *reinterpret_cast<uint32_t*>(&v[j][0]) =
                                    __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)((uint32_t)cs.idx[j] * 4u), 0, 0);
                        }
                    } else
#pragma unroll
                    for (int j = 0; j < (1 << DIM); ++j) {
                        const int voff = (int)((uint32_t)cs.idx[j] * (uint32_t)(F * sizeof(T)));
                        if constexpr (W == 1) {
                            *reinterpret_cast<uint32_t*>(&v[j][0]) = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0);
                        } else if constexpr (W == 2) {
                            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
                            *reinterpret_cast<u32x2_t*>(&v[j][0]) = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0);
                        } else if constexpr (W == 4) {
                            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                            *reinterpret_cast<u32x4_t*>(&v[j][0]) = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < (1 << DIM); ++j) fetch(j, table + (int64_t)cs.idx[j] * F);
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
        __builtin_amdgcn_wave_barrier();
        // write the tile back as one contiguous block: consecutive lanes -> consecutive dwords of feats
        const int64_t rows = (n - tile * HG_TILE) < HG_TILE ? (n - tile * HG_TILE) : HG_TILE;
        const int total = (int)rows * row_dw;
        uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(feats) + tile * HG_TILE * row_dw;
        for (int gidx = lane; gidx < total; gidx += 64) {
            const int s = row_shift >= 0 ? gidx >> row_shift : gidx / row_dw;
            const int rem = gidx - s * row_dw;
            const int l = rem / W;
            const int w = rem - l * W;
            out[gidx] = stage[(l * 65 + s) * W + w];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Backward.  Global fp32 atomics on MI355X execute memory-side at ~1.8e10 /s no matter the scope bits
// (PMC: TCC_EA0_ATOMIC == number of atomic instructions), so the levers are (1) the NUMBER of atomics and (2) not using
// them at all where a table slice fits in LDS.
//  (1) Run merge.  Consecutive samples of a ray are consecutive lanes and on every level several of them fall into the
//      same grid cell (~2 per cell at res 512, the whole wave at res 16 for the 2048-step march), and a cell fixes all
//      2^DIM corner indices.  Each wave runs a segmented (by cell id) inclusive scan over its 64 lanes for the
//      2^DIM x F corner contributions; only the LAST lane of every run emits: 36.5 instead of 256 per sample.
//  (2) Binned LDS reduction: the emit kernel writes (index, values) records into buckets of 8192 table entries per level;
//      a second kernel gives every (level, bucket) to ONE workgroup, which accumulates the bucket in LDS and adds the
//      finished slice to the gradient table with plain coalesced stores.  A bucket is stored as one fixed-capacity SLOT per
//      emitting workgroup ([bucket][tile][slot_cap] records + a count): the emitter needs no histogram pass, no global
//      reservation atomics (they are memory-side too: 2 M of them cost 0.1 ms) and no second pass over its registers - it
//      ranks a record inside its slot with one LDS counter and writes it straight away.  Records cost 8-12 B of
//      streamed traffic instead of a memory-side atomic; a full slot (capacity = 1.5 x / 4 x the no-merge expectation on
//      hashed / dense levels, at least 128) falls back to the atomic.
struct LevelList { int32_t n; int32_t lv[HG_MAX_LODS]; };
// per binned level: #buckets, slot capacity (records), first count cell, first record, table entries, split factor
struct BinLevels {
    int32_t chunks[HG_MAX_LODS]; uint32_t cap[HG_MAX_LODS]; int64_t cnt_base[HG_MAX_LODS]; int64_t rec_base[HG_MAX_LODS];
    uint32_t entries[HG_MAX_LODS]; int32_t splits[HG_MAX_LODS];
    int32_t blk_base[HG_MAX_LODS + 1];      // reduce kernel: first workgroup of every level in the flattened 1-D grid
    int32_t rank_base[HG_MAX_LODS + 1];     // emit kernel: first LDS rank counter of every level
    int64_t max_base;                       // count cell of emitting workgroup 0's largest record magnitude ([ntiles] cells)
    // Bucket of a table row.  Hashed levels and one-bucket levels: bucket = row >> chunk_shift (8192 consecutive rows).  DENSE
    // levels with several buckets (strip_magic != 0): a bucket of consecutive rows is a slab of space, a workgroup's rays cross
    // few slabs, and its records of the level land in a few (bucket, workgroup) slots - fullest slot 3-8 x the mean, which is what
    // the slot capacity (hence the scratch) has to follow, and whole buckets 3 x the mean load for the reduce kernel.  There the
    // level is cut into STRIPS of 32 rows dealt round-robin to the buckets, the deal rotated by the round number so that rows a
    // multiple of the bucket count apart (the next y row / z plane of a power-of-two grid) do not meet in one bucket:
    //   strip = row >> 5, round = strip / buckets, bucket = (strip % buckets + (round & rot_mask)) mod buckets,
    //   entry inside the bucket = round * 32 + (row & 31)            (fullest slot 1.3-2 x the mean, bucket loads within 15 %)
    uint32_t strip_magic[HG_MAX_LODS];      // ceil(2^32 / buckets), 0 = consecutive rows
    uint32_t rot_mask[HG_MAX_LODS];         // (largest power of two <= buckets) - 1
};
#define HG_STRIP_SHIFT 5
// -> bucket and entry inside it (wave-uniform branch)
static __device__ __forceinline__ void bucket_of(uint32_t idx, int chunk_shift, uint32_t magic, uint32_t buckets, uint32_t rot_mask,
                                                 uint32_t& b, uint32_t& loc) {
    if (magic == 0) {
        b = idx >> chunk_shift;
        loc = idx & ((1u << chunk_shift) - 1u);
    } else {
        const uint32_t strip = idx >> HG_STRIP_SHIFT;
        const uint32_t round = __umulhi(strip, magic);           // exact: strip < 2^18, buckets <= 1024 (bin_plan)
        uint32_t t = strip - __umul24(round, buckets) + (round & rot_mask);
        b = t >= buckets ? t - buckets : t;
        loc = (round << HG_STRIP_SHIFT) | (idx & ((1u << HG_STRIP_SHIFT) - 1u));
    }
}
// the inverse: row of entry `loc` of bucket b (may lie past the level's last row: the caller checks)
static __device__ __forceinline__ uint32_t bucket_row(uint32_t b, uint32_t loc, int chunk_shift, uint32_t magic, uint32_t buckets, uint32_t rot_mask) {
    if (magic == 0) return (b << chunk_shift) + loc;
    const uint32_t round = loc >> HG_STRIP_SHIFT;
    const uint32_t rot = round & rot_mask;
    const uint32_t r = b >= rot ? b - rot : b + buckets - rot;
    return ((round * buckets + r) << HG_STRIP_SHIFT) | (loc & ((1u << HG_STRIP_SHIFT) - 1u));
}

template <typename T, int F, int DIM, bool MERGE>
static __device__ __forceinline__ bool tail_compute(const float* c, bool live, int l, int32_t res, float hi, float hr, bool dense, uint32_t tsize,
                                                    bool pow2, int zero_from_col, const T* gp, int lane,
                                                    CornerSetup<DIM>& cs, float (&v)[1 << DIM][F]) {
    // gp -> the F gradient values of this (sample, level) (global memory or an LDS copy); read only for live samples
    float g[F];
#pragma unroll
    for (int k = 0; k < F; ++k) g[k] = 0.0f;
    if (live) {
#pragma unroll
        for (int k = 0; k < F; ++k) g[k] = (l * F + k < zero_from_col) ? Cvt<T>::to_f(gp[k]) : 0.0f;
    }
    corner_setup<DIM>(c, res, hi, hr, dense, tsize, pow2, cs);
    {
        typedef float f32x2 __attribute__((ext_vector_type(2)));     // two features per v_pk_mul_f32
        static_assert(F % 2 == 0, "feature_dim is even");
#pragma unroll
        for (int j = 0; j < (1 << DIM); ++j)
#pragma unroll
            for (int k = 0; k < F; k += 2) {
                const f32x2 gg = {g[k], g[k + 1]};
                const f32x2 r = gg * cs.coef[j];
                v[j][k] = r[0]; v[j][k + 1] = r[1];
            }
    }
    if (!MERGE) return live;
    // Two samples belong to one run iff they sit in the same cell.  Only NEIGHBOURING lanes are ever compared, so the cell
    // is carried as (x | y << 16, z) - exact for res <= 65536 (checked on the host), no integer multiplies.
    int32_t key_xy = cs.cell[0], key_z = 0;
    if constexpr (DIM > 1) key_xy |= cs.cell[1] << 16;
    if constexpr (DIM > 2) key_z = cs.cell[2];
    if (!live) key_z = -1 - lane;                          // never equal to a neighbour
    // wave_shr:1 DPP move (lane 0 keeps its own value and is forced to be a head below)
    const int32_t prev_xy = __builtin_amdgcn_update_dpp(key_xy, key_xy, 0x138, 0xf, 0xf, false);
    const int32_t prev_z = __builtin_amdgcn_update_dpp(key_z, key_z, 0x138, 0xf, 0xf, false);
    // Run-head flags of the whole wave as ONE 64-bit scalar: everything the scan needs to know about the flags (who takes
    // its predecessor's partial sum in a step, how the flags combine, which lanes are run tails) is bit arithmetic on the
    // scalar unit; the vector unit only does the value updates.
    uint64_t heads = __builtin_amdgcn_ballot_w64((key_xy != prev_xy) | (key_z != prev_z)) | 1ull;
    const uint64_t tails = (heads >> 1) | 0x8000000000000000ull;      // lane i ends a run iff lane i + 1 starts one
    // Segmented inclusive scan on the VALU (DPP), no LDS traffic: four row_shr steps inside each 16-lane row, then the
    // row totals are carried across rows with row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3).  (v, f) pairs
    // combine as (v1,f1)+(v2,f2) = (f2 ? v2 : v1+v2, f1|f2), which is associative, so the row carries compose.
    // One step: v += shifted(v) * tk with tk = 1 where the lane takes its predecessor's partial sum, else 0 - a single
    // v_fmac_f32 with the DPP shift on its first source per value (lanes whose source is outside the row / row mask keep v).
    // Written as inline asm because the compiler otherwise splits it into v_mov_dpp + packed fma + register shuffles; the
    // leading s_nop covers the VALU-write -> DPP-read hazard the assembler does not see.  (A non-finite gradient would
    // leak into neighbouring runs through 0 * inf - gradients that far gone are lost anyway.)  A step in which NO lane of
    // the wave takes anything (no run longer than the step's distance: the usual case on the fine levels, where a cell
    // holds two or three consecutive samples) is skipped with one scalar branch.
#define HG_SEG_STEP(DPPSTR, VALID, SRC_FLAGS)                                                              \
    {                                                                                                      \
        const uint64_t take = (VALID) & ~heads;                                                            \
        if (take != 0) {                                                                                   \
            float tk;                                                                                      \
            asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, %1\n\ts_nop 1" : "=v"(tk) : "s"(take));            \
            _Pragma("unroll") for (int j = 0; j < (1 << DIM); ++j)                                         \
                _Pragma("unroll") for (int k = 0; k < F; ++k)                                              \
                    asm volatile("v_fmac_f32_dpp %0, %0, %1 " DPPSTR : "+v"(v[j][k]) : "v"(tk));             \
            heads |= take & (SRC_FLAGS);                                                                   \
        }                                                                                                  \
    }
    HG_SEG_STEP("row_shr:1 row_mask:0xf bank_mask:0xf", 0xfffefffefffefffeull, heads << 1)
    HG_SEG_STEP("row_shr:2 row_mask:0xf bank_mask:0xf", 0xfffcfffcfffcfffcull, heads << 2)
    HG_SEG_STEP("row_shr:4 row_mask:0xf bank_mask:0xf", 0xfff0fff0fff0fff0ull, heads << 4)
    HG_SEG_STEP("row_shr:8 row_mask:0xf bank_mask:0xf", 0xff00ff00ff00ff00ull, heads << 8)
    HG_SEG_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf", 0xffff0000ffff0000ull,                      // rows 1 and 3
                (((heads >> 15) & 1ull) ? 0x00000000ffff0000ull : 0ull) | (((heads >> 47) & 1ull) ? 0xffff000000000000ull : 0ull))
    HG_SEG_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf", 0xffffffff00000000ull,                      // rows 2 and 3
                ((heads >> 31) & 1ull) ? 0xffffffff00000000ull : 0ull)
#undef HG_SEG_STEP
    // run tail holds the run total
    const uint64_t emitters = tails & __builtin_amdgcn_ballot_w64(live);
    uint32_t is_tail;
    asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(is_tail) : "s"(emitters));
    return is_tail != 0;
}

// direct-atomic path: wave w of a workgroup handles level levels.lv[w] for a tile of 64 samples
template <typename T, int F, int DIM, bool MERGE>
__global__ void __launch_bounds__(1024)
hashgrid_bwd_kernel(const float* __restrict__ coords, int64_t n, const T* __restrict__ grad_feats,
                    const int64_t* __restrict__ first_idx, HashLevels lv, LevelList levels, int num_lods, uint32_t tsize,
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
        for (int li = wave; li < levels.n; li += nwaves) {
            const int l = __builtin_amdgcn_readfirstlane(levels.lv[li]);
            const int32_t res = __builtin_amdgcn_readfirstlane(lv.res[l]);
            const bool dense = __builtin_amdgcn_readfirstlane(lv.dense[l]) != 0;
            CornerSetup<DIM> cs;
            float v[1 << DIM][F];
            const bool issue = tail_compute<T, F, DIM, MERGE>(c, live, l, res, lv.hi[l], lv.hr[l], dense, tsize, tsize_pow2 != 0, zero_from_col,
                                                              grad_feats + (i * num_lods + l) * F, lane, cs, v);
            if (issue) {
                const int64_t base = first_idx[l], total_rows = first_idx[num_lods];
#pragma unroll
                for (int j = 0; j < (1 << DIM); ++j) {
                    const int64_t row = base + (int64_t)(uint32_t)cs.idx[j];
                    if (row >= total_rows) continue;          // spill index past the whole table (reference: out of bounds)
                    float* p = grad_codebook + row * F;
#pragma unroll
                    for (int k = 0; k < F; ++k) atomicAdd(p + k, v[j][k]);   // global_atomic_add_f32
                }
            }
        }
    }
}

// ---- binned path for hashed levels -----------------------------------------------------------------------------------
#ifndef EM_THREADS
#define EM_THREADS 512
#endif
#ifndef EM_GROUPS
#define EM_GROUPS 2           // 64-sample groups per wave (1024-sample tiles; 4096 measured slower in the reduce)
#endif
#define EM_TILE (EM_THREADS * EM_GROUPS)     // samples per emit workgroup
#ifndef EM_MIN_WAVES
#define EM_MIN_WAVES 2          // waves per SIMD the emit kernel is register-budgeted for
#endif
// The queue emitter (the training shape) comes in two widths, two 64-sample groups per wave in both:
//   EQ_THREADS  (512:  1024-sample pieces, two workgroups per CU) - small launches: at the reference trainer's 2^18 samples the wide
//                form would leave half of the CUs without a workgroup (0.320 -> 0.335-0.346 ms per step);
//   EQ_WIDE     (1024: 2048-sample pieces, ONE workgroup per CU)  - launches of at least EQ_WIDE_MIN samples: half as many
//                (bucket, workgroup) slots, twice as full - the fullest slot of a hashed level is 1.6-2.1 x the mean instead of
//                1.9-2.6 x, the scratch the capacities add up to falls by a fifth (1.14 -> 0.91 GB at 2^21 samples), the reduce
//                kernel walks fuller chunks: pair 0.417 vs 0.420 ms in alternating runs.  (1024 threads with ONE group per wave - 1024-sample pieces - cost
//                +7 us: the per-piece set-up of a wave is then paid per 64 samples; profiles/r05_ab_scratch_geometry.txt.)
//   WISP_HG_EMIT_WIDE=0 keeps the narrow form everywhere.
#ifndef EQ_THREADS
#define EQ_THREADS 512
#endif
#define EQ_WIDE 1024
#define EQ_WIDE_MIN ((int64_t)1 << 20)
//   EQ_SMALL    (1024 threads, ONE 64-sample group per wave: 1024-sample pieces, one workgroup per CU) - launches below
//                WISP_HG_EMIT_SMALL_BELOW samples (default EQ_WIDE_MIN; 0 = never): at the reference trainer's 2^18 samples the 512-thread form makes 256
//                workgroups of one piece each - two waves per SIMD walking two groups x 15 levels one after the other, a pass whose
//                latency nothing hides; the same 256 pieces as 16 waves of one group each put four waves on a SIMD.
//                (256 threads x 2 groups - 512-sample pieces, twice the slots - was measured: pair 143 -> 201 us, the reduce
//                kernel pays for every slot.)
#define EQ_SMALL 1024
#define EQ_SMALL_GROUPS 1
#define RD_THREADS 1024
#define BIN_MAX_CHUNKS 1024
#ifndef HG_ACC_PLANES
#define HG_ACC_PLANES 1        // (0: A/B builds with the bucket accumulators interleaved by feature, as they were)
#endif
#ifndef HG_FLUSH_PAIRS
#define HG_FLUSH_PAIRS 1       // (0: A/B builds of the reduce kernel's flush without its 8-byte path, scripts/gpu_r4_k.sh)
#endif

// Records.  Generic form: { entry inside the bucket, F fp32 gradient values } = 1 + F dwords.  For two features coming
// from a 16-bit gradient tensor (the bf16 / fp16 training path) the record is packed into TWO dwords: each value keeps
// sign, exponent and 16 mantissa bits (rounded; 2^-17 relative - the values were products of a 16-bit gradient already)
// and donates its low 7 bits to the index INSIDE the bucket (14 bits; a bucket has at most 8192 entries).  A third less
// record traffic in both kernels.
template <typename T, int F> struct RecordCodec {
    static constexpr bool COMPACT = (sizeof(T) == 2 && F == 2);
    static constexpr int RW = COMPACT ? 2 : 1 + F;
    // -> the largest magnitude stored, as float bits (sign cleared): feeds the reduce kernel's fixed-point exponent
    static __device__ __forceinline__ uint32_t store(uint32_t* dst, uint32_t loc, const float (&v)[F]) {
        if constexpr (COMPACT) {
            uint2 w;
            // (round to 17 bits.  The add cannot carry out of a NaN's mantissa into the sign: COMPACT records come from 16-bit
            //  gradients, whose NaN payloads have zero low bits, and products / sums hand a NaN operand's payload on - or
            //  produce the canonical 0x7fc00000; an inf stays inf)
            w.x = ((__float_as_uint(v[0]) + 0x40u) & ~0x7fu) | (loc & 0x7fu);
            w.y = ((__float_as_uint(v[1]) + 0x40u) & ~0x7fu) | (loc >> 7);
            *reinterpret_cast<uint2*>(dst) = w;
            return max(w.x & 0x7fffff80u, w.y & 0x7fffff80u);
        } else {
            dst[0] = loc;
            uint32_t m = 0;
#pragma unroll
            for (int k = 0; k < F; ++k) { dst[1 + k] = __float_as_uint(v[k]); m = max(m, __float_as_uint(v[k]) & 0x7fffffffu); }
            return m;
        }
    }
    // -> entry index inside the bucket, values
    static __device__ __forceinline__ uint32_t load(const uint32_t (&w)[RW], float (&v)[F]) {
        if constexpr (COMPACT) {
            v[0] = __uint_as_float(w[0] & ~0x7fu);
            v[1] = __uint_as_float(w[1] & ~0x7fu);
            return (w[0] & 0x7fu) | ((w[1] & 0x7fu) << 7);
        } else {
#pragma unroll
            for (int k = 0; k < F; ++k) v[k] = __uint_as_float(w[1 + k]);
            return w[0];
        }
    }
};

// (records: see RecordCodec)
// A workgroup owns EM_TILE consecutive samples for ALL levels and slot [bucket][blockIdx.x] of every bucket.  The
// [EM_TILE x L*F] gradient rows are read from HBM once, fully coalesced, into LDS (row stride padded to an odd dword
// count: the per-level column reads are conflict free) - a (tile, level) grid re-reads every 64-byte row once per level
// (PMC: 2.3 GB fetched for 0.13 GB of gradients), which is what bounded this kernel.  After the staging barrier the waves
// run independently: every level has its own LDS rank counters, so there is no barrier inside the level loop.
template <typename T, int F, int DIM>
__global__ void __launch_bounds__(EM_THREADS, EM_MIN_WAVES)
hashgrid_bwd_emit_kernel(const float* __restrict__ coords, int64_t n, const T* __restrict__ grad_feats,
                         const int64_t* __restrict__ first_idx, HashLevels lv, LevelList levels, int num_lods,
                         uint32_t tsize, int tsize_pow2, int zero_from_col, int chunk_shift, BinLevels bins,
                         uint32_t* __restrict__ counts, uint32_t* __restrict__ records, float* __restrict__ grad_codebook) {
    constexpr int NC = 1 << DIM;
    typedef RecordCodec<T, F> Codec;
    constexpr int RW = Codec::RW;
    constexpr int GROUPS = EM_TILE / EM_THREADS;         // 64-sample groups per wave
    static_assert((F * sizeof(T)) % 4 == 0, "a (sample, level) gradient is a whole number of dwords");
    constexpr int W = (F * (int)sizeof(T)) / 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t em_smem[];
    const int total_ranks = bins.rank_base[levels.n];
    uint32_t* s_rank = em_smem;                          // [total_ranks] records written so far per (level, bucket)
    uint32_t* s_mx = em_smem + total_ranks;              // largest record magnitude of this workgroup (float bits)
    uint32_t* s_grad = em_smem + total_ranks + 1;        // [EM_TILE][rowp] gradient rows of this tile
    uint32_t mx = 0;
    if (threadIdx.x == 0) *s_mx = 0;
    const int row_dw = num_lods * W;
    const int rowp = row_dw | 1;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t ntiles = gridDim.x;
    const int64_t tile0 = (int64_t)blockIdx.x * EM_TILE;
    const int64_t total_rows = first_idx[num_lods];     // rows of the whole table
    for (int b = threadIdx.x; b < total_ranks; b += EM_THREADS) s_rank[b] = 0;
    {
        const int64_t rows = (n - tile0) < (int64_t)EM_TILE ? (n - tile0) : (int64_t)EM_TILE;
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(grad_feats) + tile0 * row_dw;
        const int total = (int)rows * row_dw;
        for (int e = threadIdx.x; e < total; e += EM_THREADS) {
            const int r = e / row_dw;
            s_grad[r * rowp + (e - r * row_dw)] = src[e];
        }
    }
    float c[GROUPS][DIM];
    bool live[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const int64_t i = tile0 + (int64_t)(wave * GROUPS + g) * 64 + lane;
        live[g] = i < n;
#pragma unroll
        for (int a = 0; a < DIM; ++a) c[g][a] = live[g] ? coords[i * DIM + a] : 0.0f;
    }
    __syncthreads();
    for (int li = 0; li < levels.n; ++li) {
        const int l = levels.lv[li];
        const int32_t res = lv.res[l];
        const bool dense = lv.dense[l] != 0;
        const uint32_t cap = bins.cap[li];
        uint32_t* rank_l = s_rank + bins.rank_base[li];
        uint32_t* __restrict__ rec_l = records + (size_t)bins.rec_base[li] * RW;
        // record offsets inside the level, 32-bit; kept opaque so that the address is ONE vector multiply per record
        // (the compiler otherwise re-associates it into (b * ntiles + blockIdx) * cap: two quarter-rate multiplies)
        const uint32_t bucket_stride = __builtin_amdgcn_readfirstlane(ntiles * cap);
        const uint32_t slot0 = __builtin_amdgcn_readfirstlane(blockIdx.x * cap);
        // rows this level owns (bounded by the plan's figure AND by the table the caller really passed)
        const int64_t rows_l = first_idx[l + 1] - first_idx[l];
        const uint32_t owned = (uint32_t)(rows_l < (int64_t)bins.entries[li] ? (rows_l < 0 ? 0 : rows_l) : (int64_t)bins.entries[li]);
        const uint32_t magic = bins.strip_magic[li], nbuckets = (uint32_t)bins.chunks[li], rot_mask = bins.rot_mask[li];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int sl = (wave * GROUPS + g) * 64 + lane;
            CornerSetup<DIM> cs;
            float v[NC][F];
            const bool issue = tail_compute<T, F, DIM, true>(c[g], live[g], l, res, lv.hi[l], lv.hr[l], dense, tsize, tsize_pow2 != 0, zero_from_col,
                                                             reinterpret_cast<const T*>(s_grad + sl * rowp + l * W), lane, cs, v);
            if (!issue) continue;
            // Rank all corners first - eight independent LDS atomics in flight - and only then write: ranking and
            // writing corner by corner put a full LDS round trip in front of every record.
            uint32_t pos[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const uint32_t idx = (uint32_t)cs.idx[j];
                // an index past the rows the level owns has no bucket (and no rank counter): straight to the atomic
                uint32_t b, loc;
                bucket_of(idx, chunk_shift, magic, nbuckets, rot_mask, b, loc);
                pos[j] = idx < owned ? atomicAdd(&rank_l[b], 1u) : 0xffffffffu;
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const uint32_t idx = (uint32_t)cs.idx[j];
                uint32_t b, loc;
                bucket_of(idx, chunk_shift, magic, nbuckets, rot_mask, b, loc);
                if (pos[j] < cap) {
                    uint32_t* dst = rec_l + (size_t)((b * bucket_stride + slot0 + pos[j]) * RW);   // < 2^32 dwords (bin_plan)
                    mx = max(mx, Codec::store(dst, loc, v[j]));
                } else {
                    // slot full, or a spill index: the memory-side atomic.  A spill lands where the reference's pointer
                    // arithmetic puts it (rows of the next level, .cu:124-161) unless that is past the whole table.
                    const int64_t row = first_idx[l] + (int64_t)idx;
                    if (row < total_rows) {
                        float* p = grad_codebook + row * F;
#pragma unroll
                        for (int k = 0; k < F; ++k) atomicAdd(p + k, v[j][k]);
                    }
                }
            }
        }
    }
    if (mx) atomicMax(s_mx, mx);
    __syncthreads();
    if (threadIdx.x == 0) counts[bins.max_base + blockIdx.x] = *s_mx;
    for (int li = 0; li < levels.n; ++li) {
        const int chunks = bins.chunks[li];
        const uint32_t cap = bins.cap[li];
        const uint32_t* rank_l = s_rank + bins.rank_base[li];
        uint32_t* __restrict__ cnt_l = counts + bins.cnt_base[li];
        for (int b = threadIdx.x; b < chunks; b += EM_THREADS) {
            const uint32_t cn = rank_l[b];
            cnt_l[(size_t)b * ntiles + blockIdx.x] = cn < cap ? cn : cap;
        }
    }
}

// Emit kernel for the training shape: two features in a 16-bit gradient tensor (compact 2-dword records), at most 16 levels.
// Same slots / counts / records as the generic kernel above, two differences:
//   * a lane keeps the whole gradient row of its sample (num_lods dwords, 64 contiguous bytes) in REGISTERS - four
//     coalesced 16-byte loads per sample, no LDS staging pass, no staging barrier;
//   * record emission is split from the scan.  Only the run tails have something to emit - 2 lanes of 64 on the coarsest
//     levels, about half of them on the finest - and ranking + addressing + packing + storing cost ~19 vector
//     instructions per corner whatever the number of live lanes: 150 of the ~400 instructions of a (64 samples, level)
//     pass.  Here the tails park (index, v0, v1) of their corners in the wave's LDS queue ([tail][corner], row stride
//     3 * corners + 1 dwords: conflict free both ways) and the wave then walks the queue with ALL lanes busy, lane = one
//     (tail, corner) entry: ceil(tails / 8) passes instead of 8.  LDS instructions of one wave execute in order, so the
//     hand-over needs no barrier.
#define EQ_MAX_ROW 16
template <typename T, int DIM, int THREADS, int GROUPS = 2>
__global__ void __launch_bounds__(THREADS, EM_MIN_WAVES)
hashgrid_bwd_emit_q_kernel(const float* __restrict__ coords, int64_t n, const T* __restrict__ grad_feats,
                           const int64_t* __restrict__ first_idx, HashLevels lv, LevelList levels, int num_lods,
                           uint32_t tsize, int tsize_pow2, int zero_from_col, int chunk_shift, BinLevels bins,
                           uint32_t* __restrict__ counts, uint32_t* __restrict__ records, float* __restrict__ grad_codebook) {
    constexpr int F = 2;
    constexpr int NC = 1 << DIM;
    typedef RecordCodec<T, F> Codec;
    static_assert(Codec::COMPACT, "two 16-bit features per level");
    constexpr int RW = Codec::RW;
    constexpr int TILE = THREADS * GROUPS;               // samples per piece (GROUPS = 64-sample groups per wave)
    constexpr int QROW = 3 * NC + 1;                     // dwords per parked tail
    extern __shared__ __attribute__((aligned(16))) uint32_t em_smem[];
    const int total_ranks = bins.rank_base[levels.n];
    uint32_t* s_rank = em_smem;
    uint32_t* s_mx = em_smem + total_ranks;              // largest record magnitude of this workgroup (float bits)
    uint32_t mx = 0;
    if (threadIdx.x == 0) *s_mx = 0;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t* s_queue = em_smem + ((total_ranks + 4) & ~3) + wave * (64 * QROW);
    const uint32_t ntiles = gridDim.x;
    const int64_t total_rows = first_idx[num_lods];
    for (int b = threadIdx.x; b < total_ranks; b += THREADS) s_rank[b] = 0;
    __syncthreads();
    // lane -> (tail, corner) of a queue pass
    const uint32_t q_lane = (uint32_t)((lane / NC) * QROW + (lane % NC) * 3);
    // The grid is capped at what the chip holds at once; a workgroup takes the TILE-sample pieces blockIdx, blockIdx + grid, ...
    // and all of them feed the same slots (the rank counters live on).  No barrier inside: the waves drift freely.
    const int64_t pieces = (n + TILE - 1) / TILE;
    for (int64_t piece = blockIdx.x; piece < pieces; piece += gridDim.x) {
        const int64_t tile0 = piece * TILE;
        float c[GROUPS][DIM];
        bool live[GROUPS];
        typedef uint32_t row_t __attribute__((ext_vector_type(EQ_MAX_ROW)));     // indexed by the (wave-uniform) level: v_movrels
        row_t grow[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int64_t i = tile0 + (int64_t)(wave * GROUPS + g) * 64 + lane;
            live[g] = i < n;
#pragma unroll
            for (int a = 0; a < DIM; ++a) c[g][a] = live[g] ? coords[i * DIM + a] : 0.0f;
            const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(grad_feats) + (live[g] ? i : 0) * num_lods;
            if (num_lods == EQ_MAX_ROW) {
#pragma unroll
                for (int q = 0; q < EQ_MAX_ROW / 4; ++q) {
                    const uint4 t = reinterpret_cast<const uint4*>(src)[q];
                    grow[g][4 * q] = t.x; grow[g][4 * q + 1] = t.y; grow[g][4 * q + 2] = t.z; grow[g][4 * q + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < EQ_MAX_ROW; ++q) grow[g][q] = q < num_lods ? src[q] : 0u;
            }
        }
        for (int li = 0; li < levels.n; ++li) {
            const int l = levels.lv[li];
            const int32_t res = lv.res[l];
            const bool dense = lv.dense[l] != 0;
            const uint32_t cap = bins.cap[li];
            uint32_t* rank_l = s_rank + bins.rank_base[li];
            uint32_t* __restrict__ rec_l = records + (size_t)bins.rec_base[li] * RW;
            const uint32_t bucket_stride = __builtin_amdgcn_readfirstlane(ntiles * cap);
            const uint32_t slot0 = __builtin_amdgcn_readfirstlane(blockIdx.x * cap);
            const int64_t base_l = first_idx[l];
            const int64_t rows_l = first_idx[l + 1] - base_l;
            const uint32_t owned = (uint32_t)(rows_l < (int64_t)bins.entries[li] ? (rows_l < 0 ? 0 : rows_l) : (int64_t)bins.entries[li]);
            const uint32_t magic = bins.strip_magic[li], nbuckets = (uint32_t)bins.chunks[li], rot_mask = bins.rot_mask[li];
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                CornerSetup<DIM> cs;
                float v[NC][F];
                uint32_t gw = grow[g][l & (EQ_MAX_ROW - 1)];
                const bool issue = tail_compute<T, F, DIM, true>(c[g], live[g], l, res, lv.hi[l], lv.hr[l], dense, tsize, tsize_pow2 != 0, zero_from_col,
                                                                 reinterpret_cast<const T*>(&gw), lane, cs, v);
                const uint64_t tails = __builtin_amdgcn_ballot_w64(issue);
                if (tails == 0) continue;
                if (issue) {
                    const uint32_t t = __builtin_amdgcn_mbcnt_hi((uint32_t)(tails >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tails, 0u));
                    uint32_t* q = s_queue + t * QROW;
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        q[3 * j] = (uint32_t)cs.idx[j];
                        q[3 * j + 1] = __float_as_uint(v[j][0]);
                        q[3 * j + 2] = __float_as_uint(v[j][1]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t total = (uint32_t)__popcll(tails) * NC;
                for (uint32_t p = 0; p < total; p += 64) {
                    if (p + lane < total) {
                        const uint32_t* q = s_queue + (p / NC) * QROW + q_lane;
                        const uint32_t idx = q[0];
                        float val[F];
                        val[0] = __uint_as_float(q[1]);
                        val[1] = __uint_as_float(q[2]);
                        uint32_t b, loc;
                        bucket_of(idx, chunk_shift, magic, nbuckets, rot_mask, b, loc);
                        // an index past the rows the level owns has no bucket (and no rank counter): straight to the atomic
                        const uint32_t pos = idx < owned ? atomicAdd(&rank_l[b], 1u) : 0xffffffffu;
                        if (pos < cap) {
                            uint32_t* dst = rec_l + (size_t)((b * bucket_stride + slot0 + pos) * RW);   // < 2^32 dwords (bin_plan)
                            mx = max(mx, Codec::store(dst, loc, val));
                        } else {
                            // slot full, or a spill index (lands where the reference's pointer arithmetic puts it, .cu:124-161,
                            // unless that is past the whole table)
                            const int64_t row = base_l + (int64_t)idx;
                            if (row < total_rows) {
                                float* pg = grad_codebook + row * F;
                                atomicAdd(pg, val[0]);
                                atomicAdd(pg + 1, val[1]);
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (mx) atomicMax(s_mx, mx);
    __syncthreads();
    if (threadIdx.x == 0) counts[bins.max_base + blockIdx.x] = *s_mx;
    for (int li = 0; li < levels.n; ++li) {
        const int chunks = bins.chunks[li];
        const uint32_t cap = bins.cap[li];
        const uint32_t* rank_l = s_rank + bins.rank_base[li];
        uint32_t* __restrict__ cnt_l = counts + bins.cnt_base[li];
        for (int b = threadIdx.x; b < chunks; b += THREADS) {
            const uint32_t cn = rank_l[b];
            cnt_l[(size_t)b * ntiles + blockIdx.x] = cn < cap ? cn : cap;
        }
    }
}

// queue + rank counters
static inline size_t queue_emitter_lds(int total_ranks, int dim, int threads) {
    return ((size_t)((total_ranks + 4) & ~3) + (size_t)(threads / 64) * 64 * (3 * (1 << dim) + 1)) * 4;
}
// Workgroups of the queue emitter one CU holds at once (registers + LDS; asked from the runtime, once per instance; the
// half and bf16 instances are the same code).  The rank counters vary a little with the level layout: 1024 is a safe figure.
template <typename T, int DIM, int THREADS, int GROUPS = 2>
static int queue_emitter_residency() {
    static const int v = [] {
        int nb = 0;
        auto eq = hashgrid_bwd_emit_q_kernel<T, DIM, THREADS, GROUPS>;
        const size_t lds = queue_emitter_lds(1024, DIM, THREADS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(eq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, eq, THREADS, lds) != hipSuccess || nb <= 0) nb = THREADS >= 1024 ? 1 : 2;
        return nb;
    }();
    return v;
}

// LDS accumulator of the reduce kernel.  LDS *float* atomics (ds_add_f32) turned out to retire roughly one lane per
// ~3 clocks per CU on gfx950 (measured: 67 M lane-adds -> 0.6 ms); integer LDS atomics do not have that problem, so the
// bucket is accumulated in 64-bit fixed point: exact, order-independent (bitwise reproducible gradients) and converted to
// fp32 once per table entry.  The binary point is set PER LAUNCH from the largest record magnitude M the emit workgroups
// reported (any gradient scale works: a GradScaler's 2^16 ... 2^24 as well as 1e-9 initial tables): with 2^(e-127) <= M <
// 2^(e-126), values are held in units of 2^-s, s = 165 - e (minus one per doubling of the launch beyond 2^21 samples), so that
// the 2^24 records of magnitude M one entry can receive from 2^21 samples still fit 63 bits
// (resolution M * 2^-39: finer than the 16-bit mantissa of a compact record by 23 bits, than fp32 accumulation by 15).
// A non-finite record (an overflowed fp16 gradient under a too-large loss scale) switches the workgroup to plain fp32
// LDS atomics, which propagate inf / NaN into the table gradient exactly like the reference's float atomics do - the
// GradScaler of an unchanged trainer then sees found_inf and skips the step.
struct AccFix64 {
    unsigned long long* acc;
    float mul;              // 2^(s - 32)
    double inv;             // 2^-s
    __device__ __forceinline__ void zero(uint32_t i) const { acc[i] = 0ull; }
    __device__ __forceinline__ void add(uint32_t i, float v) const {
        // trunc(v * 2^s) without fp64: a = |v| * 2^(s-32) is exact (power-of-two scaling) and below 2^8, so are its floor
        // (the high word) and the remainder in [0, 1) (the low word); the sign is applied to the 64-bit magnitude
        const float a = fabsf(v) * mul;
        const float h = floorf(a);
        const unsigned long long mag = ((unsigned long long)(uint32_t)h << 32) | (uint32_t)((a - h) * 4294967296.0f);
        const unsigned long long sgn = (unsigned long long)(long long)((int32_t)__float_as_uint(v) >> 31);   // 0 or ~0
        const unsigned long long q = (mag ^ sgn) - sgn;
        atomicAdd(acc + i, q);                                                                               // ds_add_u64
    }
    __device__ __forceinline__ float get(uint32_t i) const { return (float)((double)(long long)acc[i] * inv); }
};
struct AccF32 {             // the non-finite fallback
    float* acc;
    __device__ __forceinline__ void zero(uint32_t i) const { acc[i] = 0.0f; }
    __device__ __forceinline__ void add(uint32_t i, float v) const { atomicAdd(acc + i, v); }
    __device__ __forceinline__ float get(uint32_t i) const { return acc[i]; }
};

// The optimizer folded into the reduce kernel's flush (wisp_hashgrid_interpolate_bwd_adamw): a workgroup that owns its slice of
// the table (splits == 1) has the slice's complete gradient in LDS - instead of adding it to the gradient table for a separate
// optimizer pass to read back (and zero), it updates parameters and moments right there.  Pointers are indexed like
// grad_codebook (table element); shadow = bf16 copy of the parameters, or null.
struct AdamFlush {
    float* p; float* m; float* v; __hip_bfloat16* shadow;
    float lr, wd, b1, b2, eps, bc1, bc2_sqrt, gscale;
};
// levels [first_level, end_level) of the table that receive no gradient, stepped by `blocks` extra workgroups of the reduce launch
#define HG_TAIL_ROWS 8192
#define WISP_WHOLE_LEVEL ((int64_t)1 << 62)               // covered_rows[l]: all rows of level l, whatever its spacing in first_idx
struct TailLevels { int first_level, end_level, blocks; };

template <typename T, int F, typename ACC, bool ADAM>
static __device__ __forceinline__ void
hashgrid_bwd_reduce_body(const ACC A, const int64_t* __restrict__ first_idx, const LevelList& levels, int chunk_shift,
                         const BinLevels& bins, uint32_t ntiles, const uint32_t* __restrict__ counts,
                         const uint32_t* __restrict__ records, float* __restrict__ grad_codebook, int li, int b, int z,
                         const AdamFlush& ad) {
    typedef RecordCodec<T, F> Codec;
    constexpr int RW = Codec::RW;
    const int splits = bins.splits[li];
    const int l = levels.lv[li];
    const uint32_t csize = 1u << chunk_shift;
    const uint32_t cap = bins.cap[li];
    const uint32_t first = (uint32_t)b << chunk_shift;
    // the slice [first, first + lim / F) lies inside the rows level l owns: a bucket is flushed with plain read-modify-writes,
    // which is only race free while no other workgroup (of this or the next level) touches those addresses
    const int64_t rows_l = first_idx[l + 1] - first_idx[l];
    const uint32_t entries = (uint32_t)(rows_l < (int64_t)bins.entries[li] ? (rows_l < 0 ? 0 : rows_l) : (int64_t)bins.entries[li]);
    // strip-dealt buckets (BinLevels::strip_magic): entry e of the bucket is row bucket_row(e), valid while below `entries`
    const uint32_t magic = bins.strip_magic[li], nbuckets = (uint32_t)bins.chunks[li], rot_mask = bins.rot_mask[li];
    const bool strips = magic != 0;
    const uint32_t strip_rounds = (((entries + (1u << HG_STRIP_SHIFT) - 1u) >> HG_STRIP_SHIFT) + nbuckets - 1u) / nbuckets;
    const uint32_t lim = (strips ? min(strip_rounds << HG_STRIP_SHIFT, csize) : (entries > first ? min(entries - first, csize) : 0u)) * F;
    // Accumulator slot of table element x = entry * F + feature of this bucket: one PLANE per feature ([F][csize]).  With the
    // features of an entry side by side ([csize][F]) the lanes of one ds_add_u64 - random entries, one feature - can only
    // land on every F-th 8-byte slot, i.e. on a 1/F of the banks.  (Worth 3 of 420 us only: what the kernel waits for is the
    // RATE of LDS atomic instructions, ~15 clocks each per CU whatever their lanes hit - DESIGN.md 8-1.)
    auto ax = [&](uint32_t x) -> uint32_t {
#if HG_ACC_PLANES
        return (x % (uint32_t)F) * csize + x / (uint32_t)F;
#else
        return x;
#endif
    };
    for (uint32_t e = threadIdx.x; e < lim; e += RD_THREADS) A.zero(ax(e));
    __syncthreads();
    const uint32_t* __restrict__ cnt = counts + bins.cnt_base[li] + (size_t)b * ntiles;
    const uint32_t* __restrict__ src = records + ((size_t)bins.rec_base[li] + (size_t)b * ntiles * cap) * RW;
    // This workgroup takes the emitting tiles z, z + splits, ...; its waves take them 64 at a time.  A wave fetches
    // 64 counts with one load, enumerates the (slot, 64-record chunk) pairs through a prefix sum over the lanes and walks
    // them with eight record loads in flight per lane (lane = record inside the chunk).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = RD_THREADS / 64;
    constexpr int INFLIGHT = 8;
    const uint32_t my_tiles = ntiles > (uint32_t)z ? (ntiles - z + splits - 1) / splits : 0u;   // tiles of this workgroup
    // tile number k of this workgroup goes to wave k % NW, so that all waves have work even when there are few tiles
    for (uint32_t base = 0; base * NW + wave < my_tiles; base += 64) {
        const uint32_t k = (base + lane) * NW + wave;
        const uint32_t cnt_of_lane = k < my_tiles ? cnt[z + k * splits] : 0u;
        const uint32_t cc = (cnt_of_lane + 63u) >> 6;     // chunks of this lane's slot
        uint32_t inc = cc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, d, 64);
            if (lane >= d) inc += t;
        }
        const uint32_t exc = inc - cc;
        const uint32_t pairs = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        for (uint32_t p0 = 0; p0 < pairs; p0 += INFLIGHT) {
            uint32_t w[INFLIGHT][RW];
            bool act[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) {
                const uint32_t pr = p0 + u < pairs ? p0 + u : p0;
                const int t = __popcll(__ballot(exc <= pr)) - 1;                     // slot that owns pair pr (wave-uniform)
                const uint32_t chunk = pr - (uint32_t)__builtin_amdgcn_readlane((int)exc, t);
                const uint32_t n_t = (uint32_t)__builtin_amdgcn_readlane((int)cnt_of_lane, t);
                const uint32_t r = chunk * 64 + lane;
                act[u] = (p0 + u < pairs) && r < n_t;
                const uint32_t tile = z + ((base + t) * NW + wave) * splits;
                const uint32_t* rec = src + ((size_t)tile * cap + (act[u] ? r : 0u)) * RW;
#pragma unroll
                for (int q = 0; q < RW; ++q) w[u][q] = rec[q];
            }
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) {
                if (act[u]) {
                    float val[F];
                    const uint32_t e = Codec::load(w[u], val);
#pragma unroll
                    for (int kk = 0; kk < F; ++kk) {
#ifdef HG_EXP_NOATOM        // (timing experiment, scripts/gpu_r4_o.sh: the record walk without its LDS atomics - results are wrong)
                        if (e == 0xffffffffu && val[kk] == 123.0f)
#endif
                        A.add(ax(e * F + kk), val[kk]);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (strips) {
        // rows of this bucket: strips of 32 (contiguous 32 * F floats each) spread over the level; addressed from the level's first row
        float* __restrict__ dl = grad_codebook + first_idx[l] * F;
        if constexpr (ADAM) {
            if (splits == 1) {                            // (F == 2: an entry is one float2; the level starts on an even float)
                const int64_t level0 = first_idx[l] * F;
                float2* __restrict__ pg = reinterpret_cast<float2*>(dl);
                float2* __restrict__ pp = reinterpret_cast<float2*>(ad.p + level0);
                float2* __restrict__ pm = reinterpret_cast<float2*>(ad.m + level0);
                float2* __restrict__ pv = reinterpret_cast<float2*>(ad.v + level0);
                uint32_t* __restrict__ ps = ad.shadow ? reinterpret_cast<uint32_t*>(ad.shadow + level0) : nullptr;
                constexpr int PQ = 4;
                const uint32_t pairs = lim / F;
                if ((((uintptr_t)grad_codebook | (uintptr_t)ad.p | (uintptr_t)ad.m | (uintptr_t)ad.v) & 7u) != 0) {
                    for (uint32_t e = threadIdx.x; e < lim; e += RD_THREADS) {          // (a caller's oddly aligned tensors)
                        const uint32_t row = bucket_row((uint32_t)b, e / (uint32_t)F, chunk_shift, magic, nbuckets, rot_mask);
                        if (row >= entries) continue;
                        const int64_t x = level0 + (int64_t)row * F + e % (uint32_t)F;
                        const float c = grad_codebook[x];
                        float w = ad.p[x], m1 = ad.m[x], m2 = ad.v[x];
                        wisp_adamw_update(w, m1, m2, (c + A.get(ax(e))) * ad.gscale, ad.lr, ad.wd, ad.b1, ad.b2, ad.eps, ad.bc1, ad.bc2_sqrt);
                        ad.p[x] = w; ad.m[x] = m1; ad.v[x] = m2;
                        if (c != 0.0f) grad_codebook[x] = 0.0f;
                        if (ad.shadow) ad.shadow[x] = __float2bfloat16(w);
                    }
                    return;
                }
                for (uint32_t i0 = threadIdx.x; i0 < pairs; i0 += PQ * RD_THREADS) {
                    float2 cg[PQ], cp[PQ], cm[PQ], cv[PQ];
                    uint32_t row[PQ];
#pragma unroll
                    for (int q = 0; q < PQ; ++q) {
                        const uint32_t i = i0 + q * RD_THREADS;
                        row[q] = i < pairs ? bucket_row((uint32_t)b, i, chunk_shift, magic, nbuckets, rot_mask) : 0xffffffffu;
                        if (row[q] < entries) { cg[q] = pg[row[q]]; cp[q] = pp[row[q]]; cm[q] = pm[row[q]]; cv[q] = pv[row[q]]; }
                    }
#pragma unroll
                    for (int q = 0; q < PQ; ++q) {
                        const uint32_t i = i0 + q * RD_THREADS;
                        if (row[q] < entries) {
                            const bool dirty = cg[q].x != 0.0f || cg[q].y != 0.0f;
                            const float g0 = cg[q].x + A.get(ax(2 * i)), g1 = cg[q].y + A.get(ax(2 * i + 1));
                            wisp_adamw_update(cp[q].x, cm[q].x, cv[q].x, g0 * ad.gscale, ad.lr, ad.wd, ad.b1, ad.b2, ad.eps, ad.bc1, ad.bc2_sqrt);
                            wisp_adamw_update(cp[q].y, cm[q].y, cv[q].y, g1 * ad.gscale, ad.lr, ad.wd, ad.b1, ad.b2, ad.eps, ad.bc1, ad.bc2_sqrt);
                            pp[row[q]] = cp[q];
                            pm[row[q]] = cm[q];
                            pv[row[q]] = cv[q];
                            if (dirty) pg[row[q]] = make_float2(0.0f, 0.0f);
                            if (ps) {
                                const __hip_bfloat16 s0 = __float2bfloat16(cp[q].x), s1 = __float2bfloat16(cp[q].y);
                                ps[row[q]] = (uint32_t)__bfloat16_as_ushort(s0) | ((uint32_t)__bfloat16_as_ushort(s1) << 16);
                            }
                        }
                    }
                }
                return;
            }
        }
        // element e = entry * F + feature of the bucket -> table element; 0xffffffff past the level's last row
        auto elem = [&](uint32_t e) -> uint32_t {
            const uint32_t row = bucket_row((uint32_t)b, e / (uint32_t)F, chunk_shift, magic, nbuckets, rot_mask);
            return row < entries ? row * (uint32_t)F + e % (uint32_t)F : 0xffffffffu;
        };
        if (splits == 1) {
            // all of a thread's loads before its first store, as below
            constexpr int MAXE = 16;                      // 16384 accumulators / 1024 threads
            float cur[MAXE];
#pragma unroll
            for (int q = 0; q < MAXE; ++q) {
                const uint32_t e = threadIdx.x + q * RD_THREADS;
                const uint32_t x = e < lim ? elem(e) : 0xffffffffu;
                cur[q] = x != 0xffffffffu ? dl[x] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < MAXE; ++q) {
                const uint32_t e = threadIdx.x + q * RD_THREADS;
                const uint32_t x = e < lim ? elem(e) : 0xffffffffu;
                if (x != 0xffffffffu) {
                    const float a = A.get(ax(e));
                    if (a != 0.0f) dl[x] = cur[q] + a;
                }
            }
        } else {
            for (uint32_t e = threadIdx.x; e < lim; e += RD_THREADS) {
                const uint32_t x = elem(e);
                const float a = A.get(ax(e));
                if (x != 0xffffffffu && a != 0.0f) atomicAdd(dl + x, a);
            }
        }
        return;
    }
    const int64_t slice = (first_idx[l] + (int64_t)first) * F;
    float* __restrict__ dst = grad_codebook + slice;
    if constexpr (ADAM) {
        if (splits == 1) {
            // gradient = what the atomic fall-backs of the emit kernel left in the table (slot overflow; normally zero, and
            // put back to zero here like the optimizer's fused zeroing would) + this bucket's sum; AdamW's arithmetic is
            // wisp_adamw_update, shared with the flat optimizer kernels: the same bits as flush + separate pass
            float* __restrict__ pp = ad.p + slice;
            float* __restrict__ pm = ad.m + slice;
            float* __restrict__ pv = ad.v + slice;
            __hip_bfloat16* __restrict__ ps = ad.shadow ? ad.shadow + slice : nullptr;
            constexpr int PQ = 4;                       // pairs per thread and round trip (8: 99 registers, the same time)
            const uint32_t pairs = lim >> 1;
            if ((lim & 1u) == 0 && (((uintptr_t)dst | (uintptr_t)pp | (uintptr_t)pm | (uintptr_t)pv) & 7u) == 0) {
                for (uint32_t i0 = threadIdx.x; i0 < pairs; i0 += PQ * RD_THREADS) {
                    float2 cg[PQ], cp[PQ], cm[PQ], cv[PQ];
#pragma unroll
                    for (int q = 0; q < PQ; ++q) {
                        const uint32_t i = i0 + q * RD_THREADS;
                        if (i < pairs) {
                            cg[q] = reinterpret_cast<const float2*>(dst)[i]; cp[q] = reinterpret_cast<const float2*>(pp)[i];
                            cm[q] = reinterpret_cast<const float2*>(pm)[i]; cv[q] = reinterpret_cast<const float2*>(pv)[i];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < PQ; ++q) {
                        const uint32_t i = i0 + q * RD_THREADS;
                        if (i < pairs) {
                            const bool dirty = cg[q].x != 0.0f || cg[q].y != 0.0f;
                            const float g0 = cg[q].x + A.get(ax(2 * i)), g1 = cg[q].y + A.get(ax(2 * i + 1));
                            wisp_adamw_update(cp[q].x, cm[q].x, cv[q].x, g0 * ad.gscale, ad.lr, ad.wd, ad.b1, ad.b2, ad.eps, ad.bc1, ad.bc2_sqrt);
                            wisp_adamw_update(cp[q].y, cm[q].y, cv[q].y, g1 * ad.gscale, ad.lr, ad.wd, ad.b1, ad.b2, ad.eps, ad.bc1, ad.bc2_sqrt);
                            reinterpret_cast<float2*>(pp)[i] = cp[q];
                            reinterpret_cast<float2*>(pm)[i] = cm[q];
                            reinterpret_cast<float2*>(pv)[i] = cv[q];
                            if (dirty) reinterpret_cast<float2*>(dst)[i] = make_float2(0.0f, 0.0f);
                            if (ps) {
                                const __hip_bfloat16 s0 = __float2bfloat16(cp[q].x), s1 = __float2bfloat16(cp[q].y);
                                reinterpret_cast<uint32_t*>(ps)[i] = (uint32_t)__bfloat16_as_ushort(s0) | ((uint32_t)__bfloat16_as_ushort(s1) << 16);
                            }
                        }
                    }
                }
            } else {
                for (uint32_t e = threadIdx.x; e < lim; e += RD_THREADS) {
                    const float c = dst[e];
                    float x = pp[e], m1 = pm[e], m2 = pv[e];
                    wisp_adamw_update(x, m1, m2, (c + A.get(ax(e))) * ad.gscale, ad.lr, ad.wd, ad.b1, ad.b2, ad.eps, ad.bc1, ad.bc2_sqrt);
                    pp[e] = x; pm[e] = m1; pv[e] = m2;
                    if (c != 0.0f) dst[e] = 0.0f;
                    if (ps) ps[e] = __float2bfloat16(x);
                }
            }
            return;
        }
    }
    if (splits == 1) {
        // this workgroup owns the slice: plain read-modify-write, four floats per thread and ALL of a thread's loads issued
        // before its first store (one memory round trip per workgroup instead of one per element)
        constexpr int MAXQ = 4;                           // 16384 floats / (1024 threads x 4)
        const uint32_t quads = lim >> 2;
        if (quads <= MAXQ * RD_THREADS && (lim & 3u) == 0 && ((uintptr_t)dst & 15u) == 0) {
            float4 cur[MAXQ];
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                const uint32_t i = threadIdx.x + q * RD_THREADS;
                if (i < quads) cur[q] = reinterpret_cast<const float4*>(dst)[i];
            }
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                const uint32_t i = threadIdx.x + q * RD_THREADS;
                if (i < quads) {
                    float4 t = cur[q];
                    t.x += A.get(ax(4 * i)); t.y += A.get(ax(4 * i + 1));
                    t.z += A.get(ax(4 * i + 2)); t.w += A.get(ax(4 * i + 3));
                    reinterpret_cast<float4*>(dst)[i] = t;
                }
            }
        } else if (HG_FLUSH_PAIRS && (lim >> 1) <= 2 * MAXQ * RD_THREADS && (lim & 1u) == 0 && ((uintptr_t)dst & 7u) == 0) {
            // the same in pairs: a level whose first row is odd (nerf_hash.yaml: every level from the fourth on - 25^3 rows
            // precede it) starts 8 bytes off a 16-byte boundary
            const uint32_t pairs = lim >> 1;
            float2 cur[2 * MAXQ];
#pragma unroll
            for (int q = 0; q < 2 * MAXQ; ++q) {
                const uint32_t i = threadIdx.x + q * RD_THREADS;
                if (i < pairs) cur[q] = reinterpret_cast<const float2*>(dst)[i];
            }
#pragma unroll
            for (int q = 0; q < 2 * MAXQ; ++q) {
                const uint32_t i = threadIdx.x + q * RD_THREADS;
                if (i < pairs) {
                    float2 t = cur[q];
                    t.x += A.get(ax(2 * i)); t.y += A.get(ax(2 * i + 1));
                    reinterpret_cast<float2*>(dst)[i] = t;
                }
            }
        } else {
            for (uint32_t e = threadIdx.x; e < lim; e += RD_THREADS) {
                const float a = A.get(ax(e));
                if (a != 0.0f) dst[e] += a;
            }
        }
    } else {
        for (uint32_t e = threadIdx.x; e < lim; e += RD_THREADS) {
            const float a = A.get(ax(e));
            if (a != 0.0f) atomicAdd(dst + e, a);         // coarse levels are split over several workgroups
        }
    }
}

template <typename T, int F, bool ADAM>
__global__ void __launch_bounds__(RD_THREADS)
hashgrid_bwd_reduce_kernel(const int64_t* __restrict__ first_idx, LevelList levels, int chunk_shift, BinLevels bins,
                           uint32_t ntiles, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ records,
                           float* __restrict__ grad_codebook, int extra_bits, AdamFlush ad, TailLevels tail) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rd_smem[];            // [chunk entries * F] accumulators
    __shared__ uint32_t s_wave_max[RD_THREADS / 64];
    if constexpr (ADAM) {
        // The levels no gradient can reach ('cat' zeroes the columns from zero_from_col on: the FINEST level of nerf_hash.yaml) still
        // take the optimizer's step - weight decay moves them - and used to be why a separate optimizer launch streamed 2^19 rows.
        // tail.blocks extra workgroups of this launch do it instead: 8192 rows each, the gradient read from the table (zero
        // unless something else wrote there; put back to zero like the optimizer's fused zeroing), wisp_adamw_update as everywhere.
        if (blockIdx.x < (unsigned)tail.blocks) {                // FIRST in dispatch order: short streaming workgroups that are gone
            const int64_t k = (int64_t)blockIdx.x;               // before the record walks need the bandwidth (at the END of the grid
                                                                 // they lengthened the launch by their own 6 us)
            // EVERY row of the tail levels as the caller's first_idx lays them out - [first_idx[first_level], first_idx[end_level]) -
            // in pieces of HG_TAIL_ROWS dealt round-robin to the tail workgroups: the launcher sizes their number from
            // min(res^dim, T) per level, but a padded or custom table may space its levels differently, and a fixed
            // "blocks x 8192 rows from the start" walk would then step some rows twice (here and in the caller's launch) and
            // others never.  covered_rows says "the whole level" for these levels.
            const int64_t begin = first_idx[tail.first_level] * F, end = first_idx[tail.end_level] * F;
            for (int64_t lo = begin + k * (int64_t)(HG_TAIL_ROWS * F); lo < end; lo += (int64_t)tail.blocks * (HG_TAIL_ROWS * F)) {
                const int64_t hi = lo + HG_TAIL_ROWS * F < end ? lo + HG_TAIL_ROWS * F : end;
                for (int64_t e = lo + threadIdx.x; e < hi; e += RD_THREADS) {
                    const float c = grad_codebook[e];
                    float x = ad.p[e], m1 = ad.m[e], m2 = ad.v[e];
                    wisp_adamw_update(x, m1, m2, c * ad.gscale, ad.lr, ad.wd, ad.b1, ad.b2, ad.eps, ad.bc1, ad.bc2_sqrt);
                    ad.p[e] = x; ad.m[e] = m1; ad.v[e] = m2;
                    if (c != 0.0f) grad_codebook[e] = 0.0f;
                    if (ad.shadow) ad.shadow[e] = __float2bfloat16(x);
                }
            }
            return;
        }
    }
    // flattened (level, bucket, split) grid
    // heaviest first: the fine (hashed) levels carry most of the records, the cheap coarse buckets fill the tail of the grid
    const int bid = (int)(gridDim.x - 1u - blockIdx.x);          // (the tail workgroups took the first tail.blocks indices)
    int li = 0;
    while (li + 1 < levels.n && bid >= bins.blk_base[li + 1]) ++li;
    const int splits = bins.splits[li];
    const int local = bid - bins.blk_base[li];
    const int b = local / splits, z = local - b * splits;
    // largest record magnitude of the launch (float bits, sign cleared): max over the emitting workgroups' cells
    uint32_t m = 0;
    for (uint32_t t = threadIdx.x; t < ntiles; t += RD_THREADS) m = max(m, counts[bins.max_base + t]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if ((threadIdx.x & 63) == 0) s_wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_wave_max[0];
#pragma unroll
    for (int w = 1; w < RD_THREADS / 64; ++w) m = max(m, s_wave_max[w]);
    const int e = (int)(m >> 23);                          // biased exponent of the largest magnitude (wave- and block-uniform)
    if (e < 255) {
        int sh = 165 - e - extra_bits;                     // see AccFix64; extra_bits: launches of more than 2^21 samples
        if (sh > 159) sh = 159;                            // 2^(sh - 32) must stay a normal float
        AccFix64 A{reinterpret_cast<unsigned long long*>(rd_smem), __uint_as_float((uint32_t)(sh - 32 + 127) << 23),
                   __longlong_as_double((long long)(1023 - sh) << 52)};
        hashgrid_bwd_reduce_body<T, F, AccFix64, ADAM>(A, first_idx, levels, chunk_shift, bins, ntiles, counts, records, grad_codebook, li, b, z, ad);
    } else {
        AccF32 A{reinterpret_cast<float*>(rd_smem)};
        hashgrid_bwd_reduce_body<T, F, AccF32, ADAM>(A, first_idx, levels, chunk_shift, bins, ntiles, counts, records, grad_codebook, li, b, z, ad);
    }
}

static int fill_levels(const int32_t* resolutions, int num_lods, int coord_dim, int64_t tsize, HashLevels& lv) {
    for (int l = 0; l < num_lods; ++l) {
        const int32_t r = resolutions[l];
        if (r < 1) return -1;
        lv.res[l] = r;
        lv.hi[l] = (float)((double)(r - 1) - 1e-5);
        lv.hr[l] = 0.5f * (float)r;
        // hash_utils.cuh:27-29 / :75-76 -- strict '<' on int32 products (wrap-around preserved)
        const int32_t ts = (int32_t)tsize;
        const int32_t r2 = (int32_t)((uint32_t)r * (uint32_t)r);
        const int32_t r3 = (int32_t)((uint32_t)r2 * (uint32_t)r);
        bool dense = (r < ts) && (r2 < ts);
        if (coord_dim == 3) dense = dense && (r3 < ts);
        lv.dense[l] = dense ? 1 : 0;
    }
    for (int l = num_lods; l < HG_MAX_LODS; ++l) { lv.res[l] = 1; lv.dense[l] = 1; lv.hi[l] = 0.0f; lv.hr[l] = 0.5f; }
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
    const size_t wave_lds = (size_t)num_lods * 65 * W * 4;
    int waves = FW_WAVES;
    while (waves > 1 && waves * wave_lds > 64 * 1024) --waves;
    const size_t lds = waves * wave_lds;
    const int pow2 = (tsize & (tsize - 1)) == 0;
    const int row_dw = num_lods * W;
    int row_shift = -1;
    if ((row_dw & (row_dw - 1)) == 0) { row_shift = 0; while ((1 << row_shift) < row_dw) ++row_shift; }
    auto kern = hashgrid_fwd_kernel<T, F, DIM>;
    if (lds > 48 * 1024) { if (const hipError_t e = WISP_ALLOW_LDS(kern, lds)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e)); }
    const int64_t tiles = ceil_div64(n, HG_TILE);
    int64_t g = ceil_div64(tiles, waves);
    if (g > 8192) g = 8192;                               // 256 CUs x 32; grid-stride beyond that
    // every level has at most tsize rows: the whole table is below 2 GB <=> 31-bit byte offsets inside a level are safe
    // (the buffer descriptor covers 2^31 - 1 bytes from the level's first row)
    const int off32 = (uint64_t)num_lods * tsize * F * sizeof(T) < 0x7fffffffull ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)(g < 1 ? 1 : g)), dim3(waves * 64), lds, s, coords, n, (const T*)codebook,
                       first_idx, lv, num_lods, tsize, pow2, zero_from_col, row_shift, off32, (T*)feats);
    return 0;
}

static bool env_flag(const char* name, bool dflt) {
    const char* e = getenv(name);
    if (!e || !e[0]) return dflt;
    return e[0] != '0';
}
static bool bwd_merge_enabled() { static const bool v = env_flag("WISP_HG_BWD_MERGE", true); return v; }
static bool bwd_bin_enabled() { static const bool v = env_flag("WISP_HG_BWD_BIN", true); return v; }

static bool strip_buckets_enabled() { static const bool v = env_flag("WISP_HG_STRIP_BUCKETS", true); return v; }
static bool queue_emitter_enabled() { static const bool v = env_flag("WISP_HG_BWD_QUEUE", true); return v; }
// workgroups of the queue emitter the chip holds at once; 0 = no cap
static int64_t queue_emitter_grid_cap(int resident_per_cu) {
    static const int forced = [] { const char* e = getenv("WISP_HG_EMIT_WGS_PER_CU"); return e && e[0] ? atoi(e) : -1; }();
    const int per_cu = forced >= 0 ? forced : resident_per_cu;
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return per_cu > 0 ? (int64_t)per_cu * cus : 0;
}

// bin geometry shared by the workspace query and the launcher
struct BinPlan { int chunk_shift, max_chunks, max_splits, total_blocks, total_ranks, emit_threads /* queue emitter width, 0 = generic emitter */; int64_t ntiles; BinLevels bins; int64_t count_bytes, record_bytes; bool ok;
                 uint32_t base_cap[HG_MAX_LODS]; };
// cap_scale (host, indexed by LEVEL, or nullptr = 1): the caller's measured ratio of what a slot of that level really receives
// to the no-merge expectation the capacities start from - see wisp_hashgrid_bwd_slot_stats.
static BinPlan bin_plan(int64_t n, const HashLevels& lv, const LevelList& levels, int feature_dim, int64_t tsize, int dim,
                        int rec_dwords, int64_t max_emitters = 0, const float* cap_scale = nullptr, int64_t piece = EM_TILE) {
    BinPlan p{};
    const int corners = 1 << dim;
    int64_t centries = 16384 / feature_dim;               // chunk entries: 64 KiB as fp32, 128 KiB as 64-bit fixed point
    {   // (A/B switch: WISP_HG_CHUNK_ENTRIES_BELOW="<entries>:<n>" uses smaller buckets for launches below n samples)
        static const struct { int64_t entries, below; } alt = [] {
            const char* e = getenv("WISP_HG_CHUNK_ENTRIES_BELOW");
            long long a = 0, b = 0;
            if (e && sscanf(e, "%lld:%lld", &a, &b) == 2 && a >= 256 && b > 0) return decltype(alt){(int64_t)a, (int64_t)b};
            return decltype(alt){0, 0};
        }();
        if (alt.entries > 0 && n < alt.below && alt.entries < centries) centries = alt.entries;
    }
    while (((int64_t)1 << (p.chunk_shift + 1)) <= centries) ++p.chunk_shift;
    // "tile" = what ONE emitting workgroup sends: EM_TILE samples, or (queue emitter) several EM_TILE pieces when the grid is
    // capped at the number of workgroups the chip holds at once - fewer, fuller slots for the reduce kernel to walk
    // (its loads then run with all 64 lanes busy: 200 -> 160 us at 2 M samples)
    p.ntiles = ceil_div64(n, piece);
    p.emit_threads = 0;
    int64_t tile_samples = piece;
    if (max_emitters > 0 && p.ntiles > max_emitters) {
        const int64_t pieces_per_wg = ceil_div64(p.ntiles, max_emitters);
        tile_samples = pieces_per_wg * piece;
        p.ntiles = ceil_div64(p.ntiles, pieces_per_wg);       // same makespan as max_emitters workgroups, no idle slots
    }
    int64_t cnt = 0, rec = 0;
    p.ok = true;
    for (int li = 0; li < levels.n; ++li) {
        const int l = levels.lv[li];
        // Rows the level OWNS in the table: res^dim on a dense level (MultiTable: min(T, res^dim), models/grids/utils.py:48-54),
        // T on a hashed one.  A dense corner index can exceed that only when the fp32 clamp bound rounds up to res - 1
        // (res >= 258): such contributions never enter a bucket (emit sends them to the guarded atomic), so
        // no bucket ever covers rows of the next level.  The kernels clamp once more against first_idx[l + 1] - first_idx[l].
        int64_t entries = tsize;
        if (lv.dense[l]) { entries = 1; for (int a = 0; a < dim; ++a) entries *= (int64_t)lv.res[l]; }
        if (entries > tsize) entries = tsize;
        const int64_t chunks = (entries + ((int64_t)1 << p.chunk_shift) - 1) >> p.chunk_shift;
        // slot = the records one emitting tile sends to one bucket.  Starting point: the NO-merge expectation under a uniform
        // spread (the run merge removes half of the records on the finest levels and ~95 % on the coarsest, so that
        // expectation already is 2-20 x what arrives).  A hash spreads a tile's records evenly over the buckets (x 1.25 for the
        // tail); on a dense level a bucket is a slab of space and a tile's rays may favour some slabs (x 2), but no tile of
        // 1024 ray-ordered samples has ever been seen to send more than ~800 merged records to one dense level, hence the
        // 2048 ceiling.  Overflow falls back to atomics, so the bound only has to be a good guess: 2.5 GB of scratch for
        // 2 M samples at the nerf_hash shape instead of the 4.5 GB the no-merge worst case asked for.
        int64_t cap = (tile_samples * corners + chunks - 1) / chunks;
        cap = lv.dense[l] ? cap * 2 : cap + cap / 4;
        if (cap < 128) cap = 128;
        if (lv.dense[l] && cap > 2048 * (tile_samples / EM_TILE)) cap = 2048 * (tile_samples / EM_TILE);       // (2048 per 1024 samples)
        if (cap > tile_samples * corners) cap = tile_samples * corners;
        p.base_cap[li] = (uint32_t)cap;
        if (cap_scale) {
            float f = cap_scale[l];
            if (!(f > 0.0f)) f = 1.0f;
            if (f < 1.0f) { cap = (int64_t)((double)cap * f) + 1; if (cap < 32) cap = 32; }
        }
        if (max_emitters > 0) {
            // slot stride = an ODD multiple of 32 records (256 B in the compact form): the workgroups of a capped grid run
            // roughly in step and write slot b at b * stride + (a common offset) - an even multiple would put them on a
            // fraction of the memory channels
            cap = (cap + 31) / 32;
            cap = (cap | 1) * 32;
        }
        if (chunks > BIN_MAX_CHUNKS || entries > 0xffffffffLL || chunks * p.ntiles * cap * rec_dwords > 0xffffffffLL) p.ok = false;
        p.bins.chunks[li] = (int32_t)chunks;
        // dense level with several buckets: strips of 32 rows dealt to the buckets (BinLevels::strip_magic)
        p.bins.strip_magic[li] = 0; p.bins.rot_mask[li] = 0;
        if (lv.dense[l] && chunks >= 2 && strip_buckets_enabled()) {
            p.bins.strip_magic[li] = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)chunks - 1) / (uint64_t)chunks);
            uint32_t pw = 1; while (pw * 2 <= (uint32_t)chunks) pw *= 2;
            p.bins.rot_mask[li] = pw - 1;
        }
        p.bins.cap[li] = (uint32_t)cap;
        p.bins.cnt_base[li] = cnt;
        p.bins.rec_base[li] = rec;
        p.bins.entries[li] = (uint32_t)entries;
        // ~64 reduce workgroups per coarse level (atomic flush).  (Round 5 tried ONE owner per bucket from 16 buckets up, so that
        // the 50^3 and 64^3 levels could take the optimizer's step in the flush too: their single workgroups walk 512 nearly empty
        // slots each and became the launch's long pole - 0.41 -> 0.49 ms; profiles/r05_ab_single_owner_mid_levels.txt.)
        // (64 is the measured optimum at 2^21 samples: 16 / 32 / 128 / 256 workgroups per coarse level run the pair in 0.445 / 0.431 /
        //  0.426 / 0.456 ms against 0.416)
        int splits = (int)(64 / chunks);
        if (splits > p.ntiles) splits = (int)p.ntiles;
        p.bins.splits[li] = splits < 1 ? 1 : splits;
        if (chunks > p.max_chunks) p.max_chunks = (int)chunks;
        if (p.bins.splits[li] > p.max_splits) p.max_splits = p.bins.splits[li];
        p.bins.blk_base[li] = p.total_blocks;
        p.total_blocks += (int)chunks * p.bins.splits[li];
        p.bins.rank_base[li] = p.total_ranks;
        p.total_ranks += (int)chunks;
        cnt += chunks * p.ntiles;
        rec += chunks * p.ntiles * cap;
    }
    p.bins.blk_base[levels.n] = p.total_blocks;
    p.bins.rank_base[levels.n] = p.total_ranks;
    p.bins.max_base = cnt;
    cnt += p.ntiles;
    p.count_bytes = (cnt * 4 + 255) / 256 * 256;
    p.record_bytes = rec * rec_dwords * 4;
    return p;
}

static bool wide_emitter_enabled() { static const bool v = env_flag("WISP_HG_EMIT_WIDE", true); return v; }
// launches below this many samples use the one-group-per-wave emitter (EQ_SMALL): measured on the bench's own step at 2^18 / 2^19 /
// 10^6 samples per step: pair 145 -> 137 / 176 -> 171 / 249 -> 248 us (profiles/r06_ab_emit_small*.txt); WISP_HG_EMIT_SMALL_BELOW=0
// keeps the 512-thread form
static int64_t small_emitter_below() {
    static const int64_t v = [] { const char* e = getenv("WISP_HG_EMIT_SMALL_BELOW"); return e ? (int64_t)atoll(e) : EQ_WIDE_MIN; }();
    return v;
}
// THE plan of a backward launch of this shape - used by the launcher, the workspace query and the slot statistics alike: the queue
// emitter's capped grid for two 16-bit features, in its wide form from EQ_WIDE_MIN samples on
static BinPlan plan_for(int64_t n, const HashLevels& lv, const LevelList& levels, int coord_dim, int feature_dim, int dtype,
                        int64_t tsize, int num_lods, const float* cap_scale) {
    const bool compact = dtype != WISP_F32 && feature_dim == 2;
    if (compact && queue_emitter_enabled() && num_lods <= EQ_MAX_ROW) {
        const bool wide = wide_emitter_enabled() && n >= EQ_WIDE_MIN;
        const bool small = !wide && n < small_emitter_below();
        int resident;
        if (wide) resident = coord_dim == 3 ? queue_emitter_residency<__hip_bfloat16, 3, EQ_WIDE>() : queue_emitter_residency<__hip_bfloat16, 2, EQ_WIDE>();
        else if (small) resident = coord_dim == 3 ? queue_emitter_residency<__hip_bfloat16, 3, EQ_SMALL, EQ_SMALL_GROUPS>() : queue_emitter_residency<__hip_bfloat16, 2, EQ_SMALL, EQ_SMALL_GROUPS>();
        else resident = coord_dim == 3 ? queue_emitter_residency<__hip_bfloat16, 3, EQ_THREADS>() : queue_emitter_residency<__hip_bfloat16, 2, EQ_THREADS>();
        const int threads = wide ? EQ_WIDE : small ? EQ_SMALL : EQ_THREADS;
        BinPlan p = bin_plan(n, lv, levels, feature_dim, tsize, coord_dim, 2, queue_emitter_grid_cap(resident), cap_scale,
                             (int64_t)threads * (small ? EQ_SMALL_GROUPS : 2));
        p.emit_threads = small ? -EQ_SMALL : threads;              // (negative: the one-group-per-wave form)
        return p;
    }
    return bin_plan(n, lv, levels, feature_dim, tsize, coord_dim, compact ? 2 : 1 + feature_dim, 0, cap_scale);
}

template <typename T, int F, int DIM>
static int launch_bwd(const float* coords, int64_t n, const void* grad_feats, const int64_t* first_idx,
                      const HashLevels& lv, int num_lods, uint32_t tsize, int zero_from_col, float* grad_codebook,
                      void* workspace, int64_t workspace_bytes, const float* cap_scale, hipStream_t s,
                      const AdamFlush* adam = nullptr, int64_t* covered_rows = nullptr) {
    // adam / covered_rows (host, [num_lods], zeroed by the caller): the optimizer folded into the reduce kernel's flush for the
    // levels whose buckets have ONE owner; covered_rows[l] = rows at the start of level l that were updated there
    const int pow2 = (tsize & (tsize - 1)) == 0;
    // the run merge carries 16 bits per cell coordinate (tail_compute); wide features would not fit the register budget
    bool merge = bwd_merge_enabled() && (F * (1 << DIM) <= 32);
    for (int l = 0; l < num_lods; ++l)
        if (lv.res[l] > 65536) merge = false;
    LevelList active{0, {0}};
    for (int l = 0; l < num_lods; ++l)
        if (l * F < zero_from_col) active.lv[active.n++] = l;
    if (active.n == 0) return 0;
    const BinPlan plan = plan_for(n, lv, active, DIM, F, sizeof(T) == 4 ? WISP_F32 : WISP_BF16, (int64_t)tsize, num_lods, cap_scale);
    // emit kernel LDS: rank counters of every (level, bucket) + the tile's gradient rows
    const size_t em_lds = ((size_t)plan.total_ranks + 1 + (size_t)EM_TILE * ((num_lods * ((F * (int)sizeof(T)) / 4)) | 1)) * 4;
    const bool can_bin = merge && bwd_bin_enabled() && workspace && plan.ok && em_lds <= 150 * 1024 &&
                         plan.count_bytes + plan.record_bytes <= workspace_bytes && n >= 4096;
    if (!can_bin) {
        const int nw = active.n < 16 ? active.n : 16;
        if (merge)
            hipLaunchKernelGGL((hashgrid_bwd_kernel<T, F, DIM, true>), dim3(hg_grid(n)), dim3(64 * nw), 0, s, coords, n,
                               (const T*)grad_feats, first_idx, lv, active, num_lods, tsize, pow2, zero_from_col, grad_codebook);
        else
            hipLaunchKernelGGL((hashgrid_bwd_kernel<T, F, DIM, false>), dim3(hg_grid(n)), dim3(64 * nw), 0, s, coords, n,
                               (const T*)grad_feats, first_idx, lv, active, num_lods, tsize, pow2, zero_from_col, grad_codebook);
        return 0;
    }
    uint32_t* counts = (uint32_t*)workspace;              // every count cell is written by the emit kernel: no memset
    uint32_t* records = (uint32_t*)((char*)workspace + plan.count_bytes);
    bool launched = false;
    if constexpr (RecordCodec<T, F>::COMPACT) {
        if (plan.emit_threads != 0) {
            const size_t q_lds = queue_emitter_lds(plan.total_ranks, DIM, plan.emit_threads < 0 ? -plan.emit_threads : plan.emit_threads);
            // (one WISP_ALLOW_LDS per kernel instance: the grant is remembered per call site)
            if (plan.emit_threads == EQ_WIDE) {
                auto eq = hashgrid_bwd_emit_q_kernel<T, DIM, EQ_WIDE>;
                if (const hipError_t e = WISP_ALLOW_LDS(eq, q_lds)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
                hipLaunchKernelGGL(eq, dim3((unsigned)plan.ntiles), dim3(EQ_WIDE), q_lds, s,
                                   coords, n, (const T*)grad_feats, first_idx, lv, active, num_lods, tsize, pow2, zero_from_col,
                                   plan.chunk_shift, plan.bins, counts, records, grad_codebook);
            } else if (plan.emit_threads == -EQ_SMALL) {
                auto eq = hashgrid_bwd_emit_q_kernel<T, DIM, EQ_SMALL, EQ_SMALL_GROUPS>;
                if (const hipError_t e = WISP_ALLOW_LDS(eq, q_lds)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
                hipLaunchKernelGGL(eq, dim3((unsigned)plan.ntiles), dim3(EQ_SMALL), q_lds, s,
                                   coords, n, (const T*)grad_feats, first_idx, lv, active, num_lods, tsize, pow2, zero_from_col,
                                   plan.chunk_shift, plan.bins, counts, records, grad_codebook);
            } else {
                auto eq = hashgrid_bwd_emit_q_kernel<T, DIM, EQ_THREADS>;
                if (const hipError_t e = WISP_ALLOW_LDS(eq, q_lds)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
                hipLaunchKernelGGL(eq, dim3((unsigned)plan.ntiles), dim3(EQ_THREADS), q_lds, s,
                                   coords, n, (const T*)grad_feats, first_idx, lv, active, num_lods, tsize, pow2, zero_from_col,
                                   plan.chunk_shift, plan.bins, counts, records, grad_codebook);
            }
            launched = true;
        }
    }
    if (!launched) {
        auto em = hashgrid_bwd_emit_kernel<T, F, DIM>;
        if (const hipError_t e = WISP_ALLOW_LDS(em, em_lds)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
        hipLaunchKernelGGL(em, dim3((unsigned)plan.ntiles), dim3(EM_THREADS), em_lds, s,
                           coords, n, (const T*)grad_feats, first_idx, lv, active, num_lods, tsize, pow2, zero_from_col,
                           plan.chunk_shift, plan.bins, counts, records, grad_codebook);
    }
    const size_t rd_lds = ((size_t)1 << plan.chunk_shift) * F * 8;
    constexpr bool CAN_ADAM = (F == 2);                    // the fused update exists for the two-feature tables
    const bool fused = CAN_ADAM && adam != nullptr && covered_rows != nullptr;
    auto rd = hashgrid_bwd_reduce_kernel<T, F, false>;
    if constexpr (CAN_ADAM) { if (fused) rd = hashgrid_bwd_reduce_kernel<T, F, true>; }
    if (const hipError_t e = WISP_ALLOW_LDS(rd, rd_lds)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
    TailLevels tail{0, 0, 0};
    if (fused) {
        for (int li = 0; li < active.n; ++li)
            if (plan.bins.splits[li] == 1) covered_rows[active.lv[li]] = (int64_t)plan.bins.entries[li];
        // the trailing levels behind the last active one (WISP_ADAM_TAIL=0 leaves them to the caller's optimizer launch)
        static const bool fold_tail = env_flag("WISP_ADAM_TAIL", true);
        const int first_dead = active.lv[active.n - 1] + 1;
        bool contiguous = true;                                // active = 0 .. first_dead - 1 (zero_from_col cuts a suffix)
        for (int li = 0; li < active.n; ++li) contiguous = contiguous && active.lv[li] == li;
        if (fold_tail && contiguous && first_dead < num_lods) {
            int64_t rows = 0;
            for (int l = first_dead; l < num_lods; ++l) {
                int64_t entries = (int64_t)tsize;
                if (lv.dense[l]) { entries = 1; for (int a = 0; a < DIM; ++a) entries *= (int64_t)lv.res[l]; }
                if (entries > (int64_t)tsize) entries = (int64_t)tsize;
                covered_rows[l] = WISP_WHOLE_LEVEL;            // every row the caller's first_idx gives the level (it clamps)
                rows += entries;                               // (an estimate: it only sizes the number of tail workgroups)
            }
            tail = TailLevels{first_dead, num_lods, (int)ceil_div64(rows, HG_TAIL_ROWS)};
        }
    }
    // one table entry can receive 2^DIM records per sample: 2^24 at the 2^21 samples the binary point is laid out for; beyond
    // that the binary point moves up with the sample count so that no sum can leave its 63 bits
    int extra_bits = 0;
    while (((int64_t)1 << (21 + extra_bits)) < n) ++extra_bits;
    hipLaunchKernelGGL(rd, dim3(plan.total_blocks + tail.blocks), dim3(RD_THREADS), rd_lds, s, first_idx, active, plan.chunk_shift,
                       plan.bins, (uint32_t)plan.ntiles, counts, records, grad_codebook, extra_bits, fused ? *adam : AdamFlush{}, tail);
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
                                             int zero_from_col, float* grad_codebook, void* workspace,
                                             int64_t workspace_bytes, const float* level_cap_scale, wisp_stream_t stream) {
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
                grad_codebook, workspace, workspace_bytes, level_cap_scale, s)
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// The backward with torch.optim.AdamW's step on the table folded into it (MultiviewTrainStep on one GPU; base_trainer.py:205-246
// configures the optimizer, multiview_trainer.py:169-174 steps it): see AdamFlush.  covered_rows (HOST, [num_lods]) receives,
// per level, how many of its leading rows were updated in the flush; every other element of the table still has its gradient in
// grad_codebook and is the caller's to update (wisp_adamw_step_groups).  Levels that were not binned, or whose buckets are split
// over several workgroups (the coarse ones), report 0.
extern "C" int wisp_hashgrid_interpolate_bwd_adamw(const float* coords, int64_t n, int coord_dim, const void* grad_feats,
                                                   int dtype, int feature_dim, const int64_t* first_idx,
                                                   const int32_t* resolutions, int num_lods, int codebook_bitwidth,
                                                   int zero_from_col, float* grad_codebook, void* workspace,
                                                   int64_t workspace_bytes, const float* level_cap_scale, float* param,
                                                   float* exp_avg, float* exp_avg_sq, void* bf16_shadow, float lr, float beta1,
                                                   float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                                                   int64_t* covered_rows, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    WISP_REQUIRE(num_lods >= 1 && num_lods <= HG_MAX_LODS, "num_lods out of range");
    WISP_REQUIRE(covered_rows && param && exp_avg && exp_avg_sq && step >= 1, "optimizer arguments missing");
    for (int l = 0; l < num_lods; ++l) covered_rows[l] = 0;
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(coords && grad_feats && first_idx && resolutions && grad_codebook, "null pointer");
    WISP_REQUIRE(coord_dim == 2 || coord_dim == 3, "coord_dim must be 2 or 3");
    WISP_REQUIRE(codebook_bitwidth >= 1 && codebook_bitwidth <= 30, "codebook_bitwidth out of range");
    WISP_REQUIRE(dtype == WISP_F32 || dtype == WISP_F16 || dtype == WISP_BF16, "bad dtype");
    HashLevels lv;
    const int64_t tsize = (int64_t)1 << codebook_bitwidth;
    WISP_REQUIRE(fill_levels(resolutions, num_lods, coord_dim, tsize, lv) == 0, "bad resolution");
    // bias corrections exactly as wisp_adamw_step_groups computes them
    const AdamFlush ad{param, exp_avg, exp_avg_sq, (__hip_bfloat16*)bf16_shadow, lr, weight_decay, beta1, beta2, eps,
                       1.0f - powf(beta1, (float)step), sqrtf(1.0f - powf(beta2, (float)step)), grad_scale};
    hipStream_t s = (hipStream_t)stream;
    HG_DISPATCH(launch_bwd, coords, n, grad_feats, first_idx, lv, num_lods, (uint32_t)tsize, zero_from_col,
                grad_codebook, workspace, workspace_bytes, level_cap_scale, s, &ad, covered_rows)
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- gradient w.r.t. the coordinates
// hashgrid_interpolate_backward_cuda(..., require_grad_coords = true) (hashgrid_interpolate.cpp:69-100, kernel body
// hashgrid_interpolate_cuda.cu:163-196; requested through wisp/ops/grid.py:109-126 when the coordinates need a gradient).
// What the reference computes is reproduced TERM BY TERM, including what its own comments call unfinished:
//   * the upstream gradient of EVERY level is read from the first level's columns (grad_output[i * L * F + j], ".cu:165 FIX IN
//     MASTER lod_idx");
//   * the last term of the y derivative subtracts corner 6 where the trilinear formula has corner 5 (.cu:185-186);
//   * the corner differences are not multiplied by d(cell position) / d(coordinate) = res / 2;
//   * the 2-D kernel accepts the flag and writes nothing: the result stays the zero [n, 3] tensor of the ATen wrapper
//     (hashgrid_interpolate.cpp:88-90 allocates [n, 3] whatever the coordinate dimension).
// A drop-in answers what the reference answers; a caller that wants the analytic gradient differentiates the lookup itself.
// One thread per sample, levels in order, features in order, fp32 - the reference's accumulation order.  Not a hot path.
template <typename T>
__global__ void __launch_bounds__(256)
hashgrid_grad_coords_kernel(const float* __restrict__ coords, int64_t n, const T* __restrict__ grad_feats,
                            const T* __restrict__ codebook, const int64_t* __restrict__ first_idx, HashLevels lv, int num_lods,
                            int F, uint32_t tsize, int pow2, float* __restrict__ grad_coords) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c[3] = {coords[i * 3], coords[i * 3 + 1], coords[i * 3 + 2]};
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    for (int l = 0; l < num_lods; ++l) {
        CornerSetup<3> cs;
        corner_setup<3>(c, lv.res[l], lv.hi[l], lv.hr[l], lv.dense[l] != 0, tsize, pow2 != 0, cs);
        const int64_t base = first_idx[l];
        const int64_t rows = first_idx[l + 1] - base;
        const T* __restrict__ tab = codebook + base * F;
        int64_t r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { int64_t q = cs.idx[k]; r[k] = (q < rows ? q : rows - 1) * F; }     // (res >= 258: the clamp bound can round up)
        const float x_ = cs.frac[0], y_ = cs.frac[1], z_ = cs.frac[2];
        const float _x = 1.0f - x_, _y = 1.0f - y_, _z = 1.0f - z_;
        for (int j = 0; j < F; ++j) {
            const float go = (float)grad_feats[i * (int64_t)num_lods * F + j];                    // .cu:165-166: level 0's columns
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (float)tab[r[k] + j];
            gx += go * ((_y * _z) * (v[4] - v[0]) + (_y * z_) * (v[5] - v[1]) + (y_ * _z) * (v[6] - v[2]) + (y_ * z_) * (v[7] - v[3]));
            gy += go * ((_x * _z) * (v[2] - v[0]) + (_x * z_) * (v[3] - v[1]) + (x_ * _z) * (v[6] - v[4]) + (x_ * z_) * (v[7] - v[6]));   // .cu:185-186
            gz += go * ((_x * _y) * (v[1] - v[0]) + (_x * y_) * (v[3] - v[2]) + (x_ * _y) * (v[5] - v[4]) + (x_ * y_) * (v[7] - v[6]));
        }
    }
    grad_coords[i * 3] = gx; grad_coords[i * 3 + 1] = gy; grad_coords[i * 3 + 2] = gz;
}

extern "C" int wisp_hashgrid_grad_coords(const float* coords, int64_t n, int coord_dim, const void* grad_feats, const void* codebook,
                                         int dtype, int feature_dim, const int64_t* first_idx, const int32_t* resolutions,
                                         int num_lods, int codebook_bitwidth, float* grad_coords, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(coords && grad_feats && codebook && first_idx && resolutions && grad_coords, "null pointer");
    WISP_REQUIRE(coord_dim == 2 || coord_dim == 3, "coord_dim must be 2 or 3");
    WISP_REQUIRE(num_lods >= 1 && num_lods <= HG_MAX_LODS, "num_lods out of range");
    WISP_REQUIRE(feature_dim >= 1, "feature_dim out of range");
    WISP_REQUIRE(codebook_bitwidth >= 1 && codebook_bitwidth <= 30, "codebook_bitwidth out of range");
    WISP_REQUIRE(dtype == WISP_F32 || dtype == WISP_F16 || dtype == WISP_BF16, "bad dtype");
    hipStream_t s = (hipStream_t)stream;
    if (coord_dim == 2) {                                   // the reference's 2-D kernel leaves the zero tensor untouched
        if (const hipError_t e = hipMemsetAsync(grad_coords, 0, sizeof(float) * 3 * (size_t)n, s)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
        return WISP_OK;
    }
    HashLevels lv;
    const int64_t tsize = (int64_t)1 << codebook_bitwidth;
    WISP_REQUIRE(fill_levels(resolutions, num_lods, coord_dim, tsize, lv) == 0, "bad resolution");
    const dim3 grid((unsigned)ceil_div64(n, 256));
    const int pow2 = 1;                                      // tsize = 2^bitwidth
    if (dtype == WISP_F32)
        hipLaunchKernelGGL(hashgrid_grad_coords_kernel<float>, grid, dim3(256), 0, s, coords, n, (const float*)grad_feats,
                           (const float*)codebook, first_idx, lv, num_lods, feature_dim, (uint32_t)tsize, pow2, grad_coords);
    else if (dtype == WISP_F16)
        hipLaunchKernelGGL(hashgrid_grad_coords_kernel<__half>, grid, dim3(256), 0, s, coords, n, (const __half*)grad_feats,
                           (const __half*)codebook, first_idx, lv, num_lods, feature_dim, (uint32_t)tsize, pow2, grad_coords);
    else
        hipLaunchKernelGGL(hashgrid_grad_coords_kernel<__hip_bfloat16>, grid, dim3(256), 0, s, coords, n, (const __hip_bfloat16*)grad_feats,
                           (const __hip_bfloat16*)codebook, first_idx, lv, num_lods, feature_dim, (uint32_t)tsize, pow2, grad_coords);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- corner query (no blend)
// wisp._C.ops.hashgrid_query_cuda / hashgrid_query_backward_cuda (wisp/csrc/ops/hashgrid_query_cuda.cu:19-186, bound in
// bindings.cpp:31-32; Python side wisp/ops/grid.py:169-245).  Peripheral in the reference (nothing in wisp/ calls it), kept so
// that the whole `ops` surface binds.  Per level its OWN codebook tensor [2^bitwidth, F]; output [N, 8, num_lods, P, F] with
// P = 2^probe_bitwidth: the eight corner rows of every level, un-blended, each repeated for every probe slot - the reference's
// forward reads row idx for every p (its TODO), the index is taken modulo (2^bitwidth - P), corner k = dx<<2 | dy<<1 | dz
// (NOT the interpolation kernel's order - it is the same order: bit 2 = x).  Backward = scatter-add of the incoming
// gradient into those rows; the reference's fp32 path adds probe p into row idx + p, its half path into row idx (both
// reproduced; bf16 follows the half path), with packed 16-bit atomics for 16-bit tables like the reference's __half2 adds.
// One thread per (sample, level): consecutive threads are the levels of one sample, so for a fixed corner they write one
// contiguous run of num_lods * P * F elements.
struct QueryLevels {
    int32_t res[HG_MAX_LODS]; int32_t dense[HG_MAX_LODS]; float hi[HG_MAX_LODS]; float hr[HG_MAX_LODS];
    void* table[HG_MAX_LODS];
};

template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
hashgrid_query_kernel(const float* __restrict__ coords, int64_t n, QueryLevels lv, int num_lods, uint32_t mod, int pow2,
                      int probe, int F, int64_t table_rows, T* __restrict__ io) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * num_lods) return;
    const int64_t i = e / num_lods;
    const int l = (int)(e - i * num_lods);
    const float c[3] = {coords[i * 3], coords[i * 3 + 1], coords[i * 3 + 2]};
    CornerSetup<3> cs;
    corner_setup<3>(c, lv.res[l], lv.hi[l], lv.hr[l], lv.dense[l] != 0, mod, pow2 != 0, cs);
    T* __restrict__ table = reinterpret_cast<T*>(lv.table[l]);
    const int64_t corner_stride = (int64_t)num_lods * probe * F;
    T* __restrict__ base = io + i * 8 * corner_stride + (int64_t)l * probe * F;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t row = (int64_t)(uint32_t)cs.idx[k];
        // A dense level whose res^3 sits just below the modulus can produce a corner index past the table when the fp32
        // clamp bound rounds up to res - 1 (res >= 258); the reference reads / writes out of bounds there.
        // Such a corner reads as zero and receives no gradient here.
        const bool inside = row + (BWD && sizeof(T) == 4 ? probe - 1 : 0) < table_rows;
        for (int p = 0; p < probe; ++p) {
            T* __restrict__ cell = base + k * corner_stride + (int64_t)p * F;
            if (!BWD) {
                const T* __restrict__ src = table + row * F;
                for (int j = 0; j < F; ++j) cell[j] = inside ? src[j] : Cvt<T>::from_f(0.0f);
            } else if (!inside) {
                continue;
            } else if constexpr (sizeof(T) == 4) {
                float* dst = reinterpret_cast<float*>(table) + (row + p) * F;           // .cu:163 adds probe p into row idx + p
                for (int j = 0; j < F; ++j) atomicAdd(dst + j, (float)cell[j]);
            } else {
                T* dst = table + row * F;                                                // .cu:150-151: the half path does not
                for (int j = 0; j < F; j += 2) {
                    if constexpr (__is_same(T, __half)) {
                        unsafeAtomicAdd(reinterpret_cast<__half2*>(dst + j), *reinterpret_cast<const __half2*>(cell + j));
                    } else {
                        typedef short s16x2 __attribute__((ext_vector_type(2)));
                        __builtin_amdgcn_global_atomic_fadd_v2bf16(reinterpret_cast<s16x2*>(dst + j), *reinterpret_cast<const s16x2*>(cell + j));
                    }
                }
            }
        }
    }
}

static int query_levels(const int32_t* resolutions, int num_lods, int64_t mod, void* const* tables, QueryLevels& q) {
    HashLevels lv;
    if (fill_levels(resolutions, num_lods, 3, mod, lv) != 0) return -1;
    for (int l = 0; l < HG_MAX_LODS; ++l) {
        q.res[l] = lv.res[l]; q.dense[l] = lv.dense[l]; q.hi[l] = lv.hi[l]; q.hr[l] = lv.hr[l];
        q.table[l] = l < num_lods ? tables[l] : nullptr;
    }
    return 0;
}

template <bool BWD>
static int launch_query(const float* coords, int64_t n, void* const* tables, int dtype, int feature_dim, const int32_t* resolutions,
                        int num_lods, int codebook_bitwidth, int probe_bitwidth, void* io, hipStream_t s) {
    const int64_t probe = (int64_t)1 << probe_bitwidth;
    const int64_t mod = ((int64_t)1 << codebook_bitwidth) - probe;                     // .cu:34 / :113 codebook_mod
    QueryLevels q;
    if (mod < 1 || query_levels(resolutions, num_lods, mod, tables, q) != 0) return -1;
    const int pow2 = (mod & (mod - 1)) == 0;
    const dim3 grid((unsigned)ceil_div64(n * num_lods, 256)), block(256);
    if (dtype == WISP_F32)
        hipLaunchKernelGGL((hashgrid_query_kernel<float, BWD>), grid, block, 0, s, coords, n, q, num_lods, (uint32_t)mod, pow2, (int)probe, feature_dim, (int64_t)1 << codebook_bitwidth, (float*)io);
    else if (dtype == WISP_F16)
        hipLaunchKernelGGL((hashgrid_query_kernel<__half, BWD>), grid, block, 0, s, coords, n, q, num_lods, (uint32_t)mod, pow2, (int)probe, feature_dim, (int64_t)1 << codebook_bitwidth, (__half*)io);
    else
        hipLaunchKernelGGL((hashgrid_query_kernel<__hip_bfloat16, BWD>), grid, block, 0, s, coords, n, q, num_lods, (uint32_t)mod, pow2, (int)probe, feature_dim, (int64_t)1 << codebook_bitwidth, (__hip_bfloat16*)io);
    return 0;
}

static int query_check(const float* coords, int64_t n, const void* tables, int dtype, int feature_dim, const int32_t* resolutions,
                       int num_lods, int codebook_bitwidth, int probe_bitwidth, const void* io) {
    WISP_REQUIRE(n >= 0, "negative n");
    WISP_REQUIRE(num_lods >= 1 && num_lods <= HG_MAX_LODS, "num_lods out of range");
    WISP_REQUIRE(codebook_bitwidth >= 1 && codebook_bitwidth <= 30 && probe_bitwidth >= 0 && probe_bitwidth < codebook_bitwidth, "bad bitwidths");
    WISP_REQUIRE(dtype == WISP_F32 || dtype == WISP_F16 || dtype == WISP_BF16, "bad dtype");
    WISP_REQUIRE(feature_dim >= 2 && feature_dim % 2 == 0, "feature_dim must be a multiple of 2 (grid.py:174)");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(coords && tables && resolutions && io, "null pointer");
    WISP_REQUIRE(n * num_lods < ((int64_t)1 << 40), "too many (sample, level) pairs");
    return WISP_OK;
}

extern "C" int wisp_hashgrid_query_fwd(const float* coords, int64_t n, const void* const* codebooks, int dtype, int feature_dim,
                                       const int32_t* resolutions, int num_lods, int codebook_bitwidth, int probe_bitwidth,
                                       void* feats, wisp_stream_t stream) {
    if (int rc = query_check(coords, n, codebooks, dtype, feature_dim, resolutions, num_lods, codebook_bitwidth, probe_bitwidth, feats)) return rc;
    if (n == 0) return WISP_OK;
    for (int l = 0; l < num_lods; ++l) WISP_REQUIRE(codebooks[l], "null codebook");
    WISP_REQUIRE(launch_query<false>(coords, n, const_cast<void* const*>(codebooks), dtype, feature_dim, resolutions, num_lods,
                                     codebook_bitwidth, probe_bitwidth, feats, (hipStream_t)stream) == 0, "bad resolution");
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_hashgrid_query_bwd(const float* coords, int64_t n, const void* grad_feats, int dtype, int feature_dim,
                                       const int32_t* resolutions, int num_lods, int codebook_bitwidth, int probe_bitwidth,
                                       void* const* grad_codebooks, wisp_stream_t stream) {
    if (int rc = query_check(coords, n, grad_codebooks, dtype, feature_dim, resolutions, num_lods, codebook_bitwidth, probe_bitwidth, grad_feats)) return rc;
    if (n == 0) return WISP_OK;
    for (int l = 0; l < num_lods; ++l) WISP_REQUIRE(grad_codebooks[l], "null gradient table");
    WISP_REQUIRE(launch_query<true>(coords, n, grad_codebooks, dtype, feature_dim, resolutions, num_lods, codebook_bitwidth,
                                    probe_bitwidth, const_cast<void*>(grad_feats), (hipStream_t)stream) == 0, "bad resolution");
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// Diagnostic: what corner_setup - the code every hash-grid kernel shares - makes of a coordinate on one level.
template <int DIM>
__global__ void __launch_bounds__(256)
hashgrid_cells_kernel(const float* __restrict__ coords, int64_t n, int32_t res, float hi, float hr, int dense, uint32_t tsize,
                      int pow2, int32_t* __restrict__ cell, float* __restrict__ frac, int32_t* __restrict__ corner_idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) c[a] = coords[i * DIM + a];
    CornerSetup<DIM> cs;
    corner_setup<DIM>(c, res, hi, hr, dense != 0, tsize, pow2 != 0, cs);
#pragma unroll
    for (int a = 0; a < DIM; ++a) { cell[i * DIM + a] = cs.cell[a]; frac[i * DIM + a] = cs.frac[a]; }
    if (corner_idx) {
#pragma unroll
        for (int j = 0; j < (1 << DIM); ++j) corner_idx[i * (1 << DIM) + j] = cs.idx[j];
    }
}

extern "C" int wisp_hashgrid_cells(const float* coords, int64_t n, int coord_dim, int32_t resolution, int codebook_bitwidth,
                                   int32_t* cell, float* frac, int32_t* corner_idx, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0, "negative n");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(coords && cell && frac, "null pointer");
    WISP_REQUIRE(coord_dim == 2 || coord_dim == 3, "coord_dim must be 2 or 3");
    WISP_REQUIRE(codebook_bitwidth >= 1 && codebook_bitwidth <= 30, "codebook_bitwidth out of range");
    HashLevels lv;
    const int64_t tsize = (int64_t)1 << codebook_bitwidth;
    WISP_REQUIRE(fill_levels(&resolution, 1, coord_dim, tsize, lv) == 0, "bad resolution");
    const int pow2 = 1;
    const dim3 grid((unsigned)ceil_div64(n, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (coord_dim == 3)
        hipLaunchKernelGGL(hashgrid_cells_kernel<3>, grid, block, 0, s, coords, n, lv.res[0], lv.hi[0], lv.hr[0], lv.dense[0],
                           (uint32_t)tsize, pow2, cell, frac, corner_idx);
    else
        hipLaunchKernelGGL(hashgrid_cells_kernel<2>, grid, block, 0, s, coords, n, lv.res[0], lv.hi[0], lv.hr[0], lv.dense[0],
                           (uint32_t)tsize, pow2, cell, frac, corner_idx);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int64_t wisp_hashgrid_bwd_workspace_bytes(int64_t n, int coord_dim, int dtype, int feature_dim,
                                                     const int32_t* resolutions, int num_lods, int codebook_bitwidth,
                                                     const float* level_cap_scale) {
    if (n <= 0 || num_lods <= 0 || num_lods > HG_MAX_LODS || feature_dim <= 0 || !resolutions) return 0;
    if (coord_dim != 2 && coord_dim != 3) return 0;
    HashLevels lv;
    const int64_t tsize = (int64_t)1 << codebook_bitwidth;
    if (fill_levels(resolutions, num_lods, coord_dim, tsize, lv) != 0) return 0;
    LevelList all{0, {0}};
    for (int l = 0; l < num_lods; ++l) all.lv[all.n++] = l;
    if (dtype == WISP_F32 || dtype == WISP_F16 || dtype == WISP_BF16) {                          // exactly the plan a launch of this dtype uses
        const BinPlan e = plan_for(n, lv, all, coord_dim, feature_dim, dtype, tsize, num_lods, level_cap_scale);
        return e.ok ? e.count_bytes + e.record_bytes : 0;
    }
    const BinPlan p = bin_plan(n, lv, all, feature_dim, tsize, coord_dim, 1 + feature_dim, 0, level_cap_scale);   // widest record form
    int64_t bytes = p.ok ? p.count_bytes + p.record_bytes : 0;
    if (feature_dim == 2 && num_lods <= EQ_MAX_ROW && queue_emitter_enabled()) {             // the queue emitter's capped grid
        const BinPlan q = plan_for(n, lv, all, coord_dim, feature_dim, WISP_BF16, tsize, num_lods, level_cap_scale);
        if (q.ok && q.count_bytes + q.record_bytes > bytes) bytes = q.count_bytes + q.record_bytes;
    }
    return bytes;
}

// Fullest slot and total record count of every level after a binned backward: one workgroup per active level goes over the
// level's count cells ([bucket][emitting workgroup]) in the workspace the launch has just used.  stats = [max x L | sum x L].
__global__ void __launch_bounds__(1024)
hashgrid_bwd_slot_stats_kernel(LevelList levels, BinLevels bins, uint32_t ntiles, int num_lods, const uint32_t* __restrict__ counts,
                               uint32_t* __restrict__ max_fill) {
    __shared__ uint32_t s_max[16], s_sum[16];
    const int li = blockIdx.x;
    const uint32_t* __restrict__ c = counts + bins.cnt_base[li];
    const int64_t cells = (int64_t)bins.chunks[li] * ntiles;
    uint32_t m = 0, sum = 0;
    for (int64_t e = threadIdx.x; e < cells; e += blockDim.x) { const uint32_t v = c[e]; m = max(m, v); sum += v; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { m = max(m, (uint32_t)__shfl_xor((int)m, d, 64)); sum += (uint32_t)__shfl_xor((int)sum, d, 64); }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = m; s_sum[threadIdx.x >> 6] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) { m = max(m, s_max[w]); sum += s_sum[w]; }
        max_fill[levels.lv[li]] = m;
        max_fill[num_lods + levels.lv[li]] = sum;
    }
}

extern "C" int wisp_hashgrid_bwd_slot_stats(int64_t n, int coord_dim, int dtype, int feature_dim, const int32_t* resolutions,
                                            int num_lods, int codebook_bitwidth, int zero_from_col, const float* level_cap_scale,
                                            const void* workspace, int64_t workspace_bytes, uint32_t* max_fill,
                                            int32_t* cap_host, int32_t* base_cap_host, wisp_stream_t stream) {
    WISP_REQUIRE(n > 0 && resolutions && max_fill && cap_host && base_cap_host, "bad arguments");
    WISP_REQUIRE(coord_dim == 2 || coord_dim == 3, "coord_dim must be 2 or 3");
    WISP_REQUIRE(num_lods >= 1 && num_lods <= HG_MAX_LODS, "num_lods out of range");
    HashLevels lv;
    const int64_t tsize = (int64_t)1 << codebook_bitwidth;
    WISP_REQUIRE(fill_levels(resolutions, num_lods, coord_dim, tsize, lv) == 0, "bad resolution");
    LevelList active{0, {0}};
    for (int l = 0; l < num_lods; ++l)
        if (l * feature_dim < zero_from_col) active.lv[active.n++] = l;
    for (int l = 0; l < num_lods; ++l) { cap_host[l] = 0; base_cap_host[l] = 0; }
    const BinPlan plan = plan_for(n, lv, active, coord_dim, feature_dim, dtype, tsize, num_lods, level_cap_scale);
    // the launch was binned iff launch_bwd's own test said so: switches, merge-able shape, a workspace that holds this plan,
    // the emit kernel's LDS budget
    bool merge = bwd_merge_enabled() && bwd_bin_enabled() && (feature_dim * (1 << coord_dim) <= 32);
    for (int l = 0; l < num_lods; ++l)
        if (lv.res[l] > 65536) merge = false;
    const int elem = dtype == WISP_F32 ? 4 : 2;
    const size_t em_lds = ((size_t)plan.total_ranks + 1 + (size_t)EM_TILE * ((num_lods * ((feature_dim * elem) / 4)) | 1)) * 4;
    if (!merge || em_lds > 150 * 1024 || !workspace || !plan.ok || plan.count_bytes + plan.record_bytes > workspace_bytes || n < 4096 ||
        active.n == 0)
        return 1;
    for (int li = 0; li < active.n; ++li) { cap_host[active.lv[li]] = (int32_t)plan.bins.cap[li]; base_cap_host[active.lv[li]] = (int32_t)plan.base_cap[li]; }
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(max_fill, 0, sizeof(uint32_t) * 2 * num_lods, s);
    hipLaunchKernelGGL(hashgrid_bwd_slot_stats_kernel, dim3(active.n), dim3(1024), 0, s, active, plan.bins, (uint32_t)plan.ntiles,
                       num_lods, (const uint32_t*)workspace, max_fill);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}
