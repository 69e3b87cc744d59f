// Backward of the dual-octree trilinear lookups (OctreeGrid / CodebookOctreeGrid) for gfx950 - order-free by construction.
//
// What it replaces: the backward of kaolin.ops.spc.unbatched_interpolate_trilinear as autograd runs it under
// OctreeGrid._interpolate / interpolate (wisp/models/grids/octree_grid.py:130-219: one lookup per active level, then cat or
// sum) and of CodebookOctreeGrid._index_features + _interpolate (wisp/models/grids/codebook_grid.py:103-172: straight-through
// softmax one-hot over 2^bw dictionary rows per corner).  The reference scatters with float atomics (free order).
//
// Here every corner gradient is a sum of fixed-point integers, which is associative: whatever order the memory-side atomic
// units retire the adds in, the bits of the result are the same.  One call = all levels:
//   1. spc_grad_absmax      M = max |w_corner * g| over every (sample, level, corner, channel) - evaluated with the very float
//                           operations of pass 2, via max_j |w_j| * max_c |g_c| (rounding is monotone) - as an integer max of
//                           float bit patterns (a NaN / inf gradient gives an all-ones exponent and selects the float path)
//   2. spc_grad_scatter     per sample and level: w_j * g_c in float (the reference's products), consecutive samples of one
//                           cell merged in the wave by a fixed-shape segmented scan, run totals converted to
//                           round(v * 2^s) (s from M and the sample count: 8 n 2^s M < 2^62, so no sum can overflow) and added
//                           with 64-bit integer atomics into the accumulator rows of the workspace; touched rows are flagged
//   3. spc_grad_finalize /  every flagged row once: accumulator -> float -> ADDED to the gradient tensor (one writer per
//      codebook_finalize    element), accumulator and flag cleared.  The codebook variant turns the row's G = sum w g into the
//                           logits gradient  p_k (D_k . G - sum_m p_m D_m . G)  and adds scale * G to the dictionary row's
//                           own fixed-point accumulator (LDS, then workspace), which
//   4. codebook_dict_flush  converts (one small launch).
// The workspace (caller-owned, ZERO before its first use) is all zero again when the call returns - except its 64-byte header,
// which every call resets itself.
#include "wisp_common.h"

#define SG_MAX_LODS 16
#define SG_THREADS 128
#ifndef SG_LINK_TAILS
#define SG_LINK_TAILS 8            // a wave links corners across tails when it has more tails than this (see spc_grad_scatter_merge_kernel)
#endif
#define CB_MAX_K 256
#define CB_MAX_F 16

struct SgLods {
    float* grad[SG_MAX_LODS];            // per-level gradient tensor ([rows_l, C] features or [rows_l, K] logits)
    const float* logits[SG_MAX_LODS];    // codebook only
    const float* dict[SG_MAX_LODS];      // codebook only
    float* grad_dict[SG_MAX_LODS];       // codebook only
    int64_t base[SG_MAX_LODS + 1];       // first accumulator row of every level
    int32_t level[SG_MAX_LODS];
};

struct SgHeader { uint32_t absmax_bits; uint32_t pad[15]; };

static inline int sg_stride(int channels) {
    if (channels <= 8) { int s = 1; while (s < channels) s <<= 1; return s; }
    return (channels + 7) / 8 * 8;
}
static inline int64_t sg_round64(int64_t v) { return (v + 63) / 64 * 64; }

// ---- fixed-point scale: values are bounded by M < 2^(E+1); `adds` contributions per accumulator at most
struct SgScale { double to_fix, to_float; int finite; int zero; };
static __device__ __forceinline__ SgScale sg_scale(uint32_t absmax_bits, int clog) {
    SgScale s;
    const int e = (int)(absmax_bits >> 23);
    s.finite = e < 0xf8;                  // (a sum of 64 such values must stay finite as a float too)
    s.zero = absmax_bits == 0;
    const int E = (e < 1 ? 1 : e) - 127;
    const int sh = 60 - clog - E;                                       // 2^clog adds of < 2^(E+1) each stay below 2^61
    s.to_fix = __longlong_as_double((long long)(sh + 1023) << 52);
    s.to_float = __longlong_as_double((long long)(1023 - sh) << 52);
    return s;
}

static __device__ __forceinline__ void sg_coeffs(const float* __restrict__ c, const int16_t* __restrict__ pt, int level,
                                                 float (&w)[8]) {
    // (the statements of trilinear_coeffs in spc_interp.hip: forward and backward must see the same weights)
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

template <int CTRL> static __device__ __forceinline__ float sg_dpp(float v) {      // v of the lane the DPP control names
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int CTRL> static __device__ __forceinline__ int sg_dpp_i(int v, int fill) {   // lanes without a source lane get `fill`
    return __builtin_amdgcn_update_dpp(fill, v, CTRL, 0xf, 0xf, false);
}

static __device__ __forceinline__ uint32_t sg_wave_umax(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------- pass 1: magnitude bound
template <typename I>
__global__ void __launch_bounds__(256)
spc_grad_absmax_kernel(const float* __restrict__ coords, const I* __restrict__ cells, int64_t cell_stride, int spv,
                       const int16_t* __restrict__ points, SgLods ml, const float* __restrict__ grad_out, int64_t n,
                       int num_lods, int channels, int sum, SgHeader* __restrict__ hdr) {
    const int out_row = sum ? channels : num_lods * channels;
    uint32_t best = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* g = grad_out + i * out_row;
        uint32_t gsum = 0;
        if (sum)
            for (int c = 0; c < channels; ++c) { const uint32_t b = __float_as_uint(g[c]) & 0x7fffffffu; gsum = b > gsum ? b : gsum; }
        const I* ch = cells + (spv == 1 ? i : i / spv) * cell_stride;
        for (int l = 0; l < num_lods; ++l) {
            const int64_t p = (int64_t)ch[l];
            if (p < 0) continue;
            uint32_t gm = gsum;
            if (!sum) {
                gm = 0;
                for (int c = 0; c < channels; ++c) {
                    const uint32_t b = __float_as_uint(g[l * channels + c]) & 0x7fffffffu;
                    gm = b > gm ? b : gm;
                }
            }
            const float res = (float)(1 << ml.level[l]);
            float wm = 1.0f;
            bool first = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float f = res * (0.5f * coords[i * 3 + a] + 0.5f) - (float)points[p * 3 + a];
                const float m = fmaxf(fabsf(f), fabsf(1.0f - f));
                wm = first ? m : wm * m;                                // (a * b) * c, as the weights are formed
                first = false;
            }
            // |w_j g_c| <= fl(max_j |w_j| * max_c |g_c|): both factors are attained and float rounding is monotone
            const uint32_t b = __float_as_uint(wm * __uint_as_float(gm)) & 0x7fffffffu;
            const uint32_t nb = (gm >= 0x7f800000u || !(wm == wm)) ? 0x7fc00000u : b;       // non-finite anywhere: say so
            best = nb > best ? nb : best;
        }
    }
    best = sg_wave_umax(best);
    // (32 k waves hammering one address cost 0.17 ms; a wave whose maximum is already covered has nothing to say - the plain read
    //  may be stale, which only costs an atomic that changes nothing)
    if ((threadIdx.x & 63) == 0 && best > __atomic_load_n(&hdr->absmax_bits, __ATOMIC_RELAXED)) atomicMax(&hdr->absmax_bits, best);
}

// ---------------------------------------------------------------------------------------------- pass 2: scatter
// Small channel counts (<= 8: nerf_octree / nerf_codebook have 5): one thread owns one sample and keeps its 8 x F products
// in registers.  Consecutive samples of a ray share the cell (16 per cell in the 'voxel' march, long runs on the coarse levels
// of any march), so the wave first adds up each run with a segmented scan - same lanes, same shape, same float adds every
// time - and only the run tails go to memory: all lanes walk the (tail, corner, channel) items, channel fastest, so that the
// F adds of a row sit in neighbouring lanes of one atomic instruction.
template <int F, typename I>
__global__ void __launch_bounds__(SG_THREADS)
spc_grad_scatter_merge_kernel(const float* __restrict__ coords, const I* __restrict__ cells, int64_t cell_stride, int spv,
                              const int16_t* __restrict__ points, const int32_t* __restrict__ trinkets, SgLods ml,
                              const float* __restrict__ grad_out, int64_t n, int num_lods, int sum, int clog, int stride,
                              int direct_stride, const SgHeader* __restrict__ hdr, uint8_t* __restrict__ flags,
                              long long* __restrict__ acc) {
    __shared__ float s_val[SG_THREADS / 64][64][8 * F];
    __shared__ int32_t s_row[SG_THREADS / 64][64][8];
    __shared__ int32_t s_info[SG_THREADS / 64][64];
    __shared__ int8_t s_fwd[SG_THREADS / 64][64 * 8];    // (bytes: the slice must not cost the kernel a workgroup per CU)
    __shared__ int8_t s_bwd[SG_THREADS / 64][64 * 8];
    const SgScale sc = sg_scale(hdr->absmax_bits, clog);
    if (sc.zero) return;                                                 // every product is zero: nothing to add
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool in = i < n;
    const int out_row = sum ? F : num_lods * F;
    float c[3] = {0.0f, 0.0f, 0.0f}, g[F];
#pragma unroll
    for (int f = 0; f < F; ++f) g[f] = 0.0f;
    if (in) {
        c[0] = coords[i * 3]; c[1] = coords[i * 3 + 1]; c[2] = coords[i * 3 + 2];
        if (sum)
#pragma unroll
            for (int f = 0; f < F; ++f) g[f] = grad_out[i * out_row + f];
    }
    const I* ch = cells + (in ? (spv == 1 ? i : i / spv) * cell_stride : 0);
    for (int l = 0; l < num_lods; ++l) {
        const int64_t p = in ? (int64_t)ch[l] : -1;
        float v[8][F];
        if (p >= 0) {
            float w[8];
            sg_coeffs(c, points + p * 3, ml.level[l], w);
            if (!sum)
#pragma unroll
                for (int f = 0; f < F; ++f) g[f] = grad_out[i * out_row + l * F + f];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int f = 0; f < F; ++f) v[j][f] = w[j] * g[f];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int f = 0; f < F; ++f) v[j][f] = 0.0f;
        }
        // Runs of consecutive samples in one cell (16 per cell in the 'voxel' march, a few in the 'ray' march, long ones on the
        // coarse levels) are added up before anything goes to memory.  Inside every 16-lane row: a segmented inclusive scan with
        // DPP row shifts - no LDS traffic; per step one predicate for all 8 F values, applied as a 0 / 1 factor (x + t * 1 rounds
        // once, like the add; x + t * 0 is x).  A run that continues across a row boundary leaves a partial total at lane 15;
        // it is added to the run's final total when the totals are written out (the corner links below).  Same lanes, same
        // order of adds every time: the float part of the sum is repeatable, the rest is integer.
        const int lane16 = lane & 15;
        const int key = p >= 0 ? (int)p : -1 - lane;                      // (invalid lanes are their own run)
        const int prev = __shfl_up(key, 1, 64), next = __shfl_down(key, 1, 64);
        const bool joins = lane > 0 && prev == key;                       // continues the run of the lane before
        int head = (lane16 == 0 || !joins) ? 1 : 0;
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const float m = head ? 0.0f : 1.0f;
            int hp;
            if (step == 0) hp = sg_dpp_i<0x111>(head, 1); else if (step == 1) hp = sg_dpp_i<0x112>(head, 1);
            else if (step == 2) hp = sg_dpp_i<0x114>(head, 1); else hp = sg_dpp_i<0x118>(head, 1);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    float t;
                    if (step == 0) t = sg_dpp<0x111>(v[j][f]); else if (step == 1) t = sg_dpp<0x112>(v[j][f]);      // row_shr:1, :2
                    else if (step == 2) t = sg_dpp<0x114>(v[j][f]); else t = sg_dpp<0x118>(v[j][f]);                // row_shr:4, :8
                    v[j][f] = __builtin_fmaf(t, m, v[j][f]);
                }
            head |= hp;
        }
#ifdef SG_EXP_FEWTAILS
        const bool tail = p >= 0 && lane == 63;
#else
        const bool tail = p >= 0 && (lane16 == 15 || next != key);
#endif
        const uint64_t tmask = __ballot(tail);
        if (tmask == 0) continue;                                         // (wave-uniform)
        float (*sv)[8 * F] = s_val[wv];
        int32_t (*sr)[8] = s_row[wv];
        int8_t* fwd = s_fwd[wv];
        int8_t* bwd = s_bwd[wv];
        const int ntails = __popcll(tmask);
        // Consecutive tails often name the same table row: the two halves of a run cut by a row boundary (all eight corners: `carry`
        // / `dropped`, always honoured) and neighbouring cells of a ray, which share a face (four corners).  A corner whose row
        // reappears in the NEXT tail can hand its total on instead of going to memory, and the last tail of such a chain adds the
        // chain up, in order: a third fewer atomic requests - which is what bounds this kernel when runs are short ('ray' march,
        // 5.7 samples per run: 460 -> 383 us), and costs more than it saves when a wave has a handful of tails ('voxel' march, 16
        // samples per cell: 222 -> 264 us).  The wave decides by its tail count.
        const bool links = ntails > SG_LINK_TAILS;
        const uint64_t cont = __ballot(joins && lane16 == 0);            // bit 16 r: row r starts inside the run row r-1 ended with
        if (tail) {
            const int rank = __popcll(tmask & ((1ull << lane) - 1ull));
            const int row0 = lane & ~15;
            const bool first_in_row = ((tmask >> row0) & ((1ull << lane16) - 1ull)) == 0;
            const bool carry = first_in_row && ((cont >> row0) & 1ull);                       // add the total parked by the tail before
            const bool dropped = lane16 == 15 && lane < 63 && ((cont >> (row0 + 16)) & 1ull);  // a partial total: parked, not written out
            s_info[wv][rank] = (carry ? 1 : 0) | (dropped ? 2 : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sr[rank][j] = trinkets[p * 8 + j];
#pragma unroll
                for (int f = 0; f < F; ++f) sv[rank][j * F + f] = v[j][f];
            }
        }
        if (links)
            for (int e = lane; e < ntails * 8; e += 64) { fwd[e] = -1; bwd[e] = -1; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (links) {
            for (int e = lane; e < (ntails - 1) * 8; e += 64) {
                const int t = e >> 3;
                const int32_t r = sr[t][e & 7];
                int hit = -1;
#pragma unroll
                for (int q = 0; q < 8; ++q) hit = sr[t + 1][q] == r ? q : hit;       // (the rows of one cell are distinct)
                fwd[e] = (int8_t)hit;
                if (hit >= 0) bwd[(t + 1) * 8 + hit] = (int8_t)(e & 7);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // all lanes walk the (tail, corner, channel) items, channel fastest: the F adds of a row sit in neighbouring lanes of ONE
        // atomic instruction (the memory side takes them as one request per 64-byte line)
        const int items = ntails * 8 * F;
        const int64_t base = ml.base[l];
        float* gd = ml.grad[l];
        for (int it = lane; it < items; it += 64) {
            const int t = it / (8 * F), rem = it - t * (8 * F);
            const int j = rem / F, f = rem - j * F;
            float total = sv[t][rem];
            if (links) {
                if (fwd[t * 8 + j] >= 0) continue;                        // handed on to the next tail
                for (int k = t, jj = j; k > 0;) {
                    const int b = bwd[k * 8 + jj];
                    if (b < 0) break;
                    --k; jj = b;
                    total += sv[k][jj * F + f];
                }
            } else {
                int info = s_info[wv][t];
                if (info & 2) continue;
                for (int k = t; info & 1;) { --k; total += sv[k][rem]; info = s_info[wv][k]; }
            }
            const int32_t crow = sr[t][j];
            if (sc.finite) {
                const int64_t row = base + crow;
                const long long q = __double2ll_rn((double)total * sc.to_fix);
#ifdef SG_EXP_NOATOMIC
                if (q == 0x7fffffffffffffffll) acc[row * stride + f] = q;
#else
                atomicAdd(reinterpret_cast<unsigned long long*>(acc + row * stride + f), (unsigned long long)q);
#endif
                if (flags && f == 0) flags[row] = 1;
            } else {
                // a non-finite (or nearly overflowing) gradient: plain float atomics straight into the gradient tensor, so that
                // inf / NaN arrive where the reference's atomics would put them (the optimizer step is skipped anyway; the 0 / 1
                // factors of the scan may have turned further lanes into NaN, which changes nothing about that)
                atomicAdd(gd + (int64_t)crow * direct_stride + f, total);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                  // the slice is rewritten by the next level
    }
}

// Any channel count (NGLOD: 16): consecutive lanes own consecutive channels of one sample, one 64-bit add per product.
template <typename I>
__global__ void __launch_bounds__(256)
spc_grad_scatter_wide_kernel(const float* __restrict__ coords, const I* __restrict__ cells, int64_t cell_stride, int spv,
                             const int16_t* __restrict__ points, const int32_t* __restrict__ trinkets, SgLods ml,
                             const float* __restrict__ grad_out, int64_t n, int num_lods, int channels, int sum, int clog,
                             int stride, int direct_stride, const SgHeader* __restrict__ hdr, uint8_t* __restrict__ flags,
                             long long* __restrict__ acc, int split_lods) {
    const SgScale sc = sg_scale(hdr->absmax_bits, clog);
    if (sc.zero) return;
    const int cpt = channels <= 64 ? channels : 64;
    const int rows_per_block = blockDim.x / cpt;
    const int ch0 = threadIdx.x % cpt;
    if ((int)threadIdx.x / cpt >= rows_per_block) return;
    const int64_t step = (int64_t)gridDim.x * rows_per_block;
    const int out_row = sum ? channels : num_lods * channels;
    const int64_t units = split_lods ? n * num_lods : n;                 // split_lods: one (sample, level) per lane group
    for (int64_t u = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / cpt; u < units; u += step) {
        const int64_t i = split_lods ? u / num_lods : u;
        const int l0 = split_lods ? (int)(u - i * num_lods) : 0, l1 = split_lods ? l0 + 1 : num_lods;
        const I* ch = cells + (spv == 1 ? i : i / spv) * cell_stride;
        for (int l = l0; l < l1; ++l) {
            const int64_t p = (int64_t)ch[l];
            if (p < 0) continue;
            float w[8];
            sg_coeffs(coords + i * 3, points + p * 3, ml.level[l], w);
            const int32_t* tr = trinkets + p * 8;
            for (int c = ch0; c < channels; c += cpt) {
                const float g = grad_out[i * out_row + (sum ? 0 : l * channels) + c];
                if (sc.finite) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int64_t row = ml.base[l] + tr[j];
                        const long long q = __double2ll_rn((double)(g * w[j]) * sc.to_fix);
                        atomicAdd(reinterpret_cast<unsigned long long*>(acc + row * stride + c), (unsigned long long)q);
                        if (flags && c == 0) flags[row] = 1;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) atomicAdd(ml.grad[l] + (int64_t)tr[j] * direct_stride + c, g * w[j]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- pass 3: accumulators -> gradients
static __device__ __forceinline__ int sg_level_of(const SgLods& ml, int num_lods, int64_t row) {
    int l = 0;
    while (l + 1 < num_lods && row >= ml.base[l + 1]) ++l;
    return l;
}

__global__ void __launch_bounds__(256)
spc_grad_finalize_kernel(SgLods ml, int num_lods, int channels, int stride, int lpr /* lanes per row: power of two <= 64 */,
                         int clog, const SgHeader* __restrict__ hdr, uint8_t* __restrict__ flags, long long* __restrict__ acc) {
    const SgScale sc = sg_scale(hdr->absmax_bits, clog);
    const int64_t total = ml.base[num_lods];
    const int rpb = blockDim.x / lpr, lane = threadIdx.x % lpr;
    for (int64_t row = (int64_t)blockIdx.x * rpb + threadIdx.x / lpr; row < total; row += (int64_t)gridDim.x * rpb) {
        if (flags && !flags[row]) continue;                              // (all lanes of a row sit in one wave: read before the clear)
        const int l = sg_level_of(ml, num_lods, row);
        float* gd = ml.grad[l] + (row - ml.base[l]) * channels;
        long long* a = acc + row * stride;
        for (int c = lane; c < channels; c += lpr) {
            const long long q = a[c];
            if (q != 0) { gd[c] += (float)((double)q * sc.to_float); a[c] = 0; }
        }
        if (flags && lane == 0) flags[row] = 0;
    }
}

// Codebook: one thread per logits row.  d logits[row, k] = p_k (D_k . G - sum_m p_m D_m . G); d dictionary[argmax] += scale G
// with scale = (1 - p) + p, the forward value of the straight-through key (codebook_grid.py:117-125).
template <int KR>
__global__ void __launch_bounds__(256)
codebook_grad_finalize_kernel(SgLods ml, int lod_begin, int lod_end, int K, int F, int stride, int clog,
                              const SgHeader* __restrict__ hdr, uint8_t* __restrict__ flags, long long* __restrict__ acc,
                              long long* __restrict__ dict_acc) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    const int nl = lod_end - lod_begin;
    long long* s_gdict = reinterpret_cast<long long*>(s_raw);            // [nl * K * F] fixed-point dictionary gradient
    float* s_dict = reinterpret_cast<float*>(s_gdict + (size_t)nl * K * F);   // [nl * K * F] dictionaries
    for (int e = threadIdx.x; e < nl * K * F; e += blockDim.x) {
        s_gdict[e] = 0;
        s_dict[e] = ml.dict[lod_begin + e / (K * F)][e % (K * F)];
    }
    __syncthreads();
    const SgScale sc = sg_scale(hdr->absmax_bits, clog);
    const int64_t first = ml.base[lod_begin], last = ml.base[lod_end];
    for (int64_t row = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < last; row += (int64_t)gridDim.x * blockDim.x) {
        long long* a = acc + row * stride;
        if (flags) {
            if (!flags[row]) continue;
            flags[row] = 0;
        } else {
            bool any = false;
#pragma unroll
            for (int f = 0; f < CB_MAX_F; ++f) if (f < F) any |= a[f] != 0;
            if (!any) continue;                                          // untouched (or cancelled to exactly zero: nothing to add)
        }
        int l = lod_begin;
        while (l + 1 < lod_end && row >= ml.base[l + 1]) ++l;
        const int64_t r = row - ml.base[l];
        long long q[CB_MAX_F];                                           // (constant trip counts + guards: these stay in registers)
        float G[CB_MAX_F];
#pragma unroll
        for (int f = 0; f < CB_MAX_F; ++f) {
            q[f] = 0; G[f] = 0.0f;
            if (f < F) { q[f] = a[f]; a[f] = 0; G[f] = (float)((double)q[f] * sc.to_float); }
        }
        const float* lrow = ml.logits[l] + r * K;
        float* grow = ml.grad[l] + r * K;
        const float* D = s_dict + (size_t)(l - lod_begin) * K * F;
        int best = 0;
        float inv;
        if (KR > 0) {
            // the row in registers (K <= KR, a multiple of four: 16-byte loads), every exponential and dictionary product once
            float x[KR > 0 ? KR : 1], e[KR > 0 ? KR : 1];
#pragma unroll
            for (int k4 = 0; k4 < KR; k4 += 4)
                if (k4 < K) {
                    const float4 t = *reinterpret_cast<const float4*>(lrow + k4);
                    x[k4] = t.x; x[k4 + 1] = t.y; x[k4 + 2] = t.z; x[k4 + 3] = t.w;
                }
            float mx = x[0];
#pragma unroll
            for (int k = 1; k < KR; ++k) if (k < K && x[k] > mx) { mx = x[k]; best = k; }    // first max wins (torch.max)
            float denom = 0.0f;
#pragma unroll
            for (int k = 0; k < KR; ++k) if (k < K) { e[k] = expf(x[k] - mx); denom += e[k]; }
            inv = 1.0f / denom;
            // argmax of the softmax VALUES, first index on ties (codebook_softmax_pick, wisp_common.h), on the registers at hand
#pragma unroll
            for (int k = KR - 1; k >= 0; --k) if (k < best && e[k] * inv == inv) best = k;
            float dot = 0.0f;
#pragma unroll
            for (int k = 0; k < KR; ++k)
                if (k < K) {
                    float dk = 0.0f;
#pragma unroll
                    for (int f = 0; f < CB_MAX_F; ++f) if (f < F) dk += D[k * F + f] * G[f];
                    x[k] = dk;
                    dot += e[k] * inv * dk;
                }
#pragma unroll
            for (int k4 = 0; k4 < KR; k4 += 4)
                if (k4 < K) {
                    float4 t = *reinterpret_cast<float4*>(grow + k4);
                    t.x += e[k4] * inv * (x[k4] - dot); t.y += e[k4 + 1] * inv * (x[k4 + 1] - dot);
                    t.z += e[k4 + 2] * inv * (x[k4 + 2] - dot); t.w += e[k4 + 3] * inv * (x[k4 + 3] - dot);
                    *reinterpret_cast<float4*>(grow + k4) = t;
                }
        } else {
            float mx = lrow[0];
            for (int k = 1; k < K; ++k) { const float x = lrow[k]; if (x > mx) { mx = x; best = k; } }   // first max wins (torch.max)
            float denom = 0.0f;
            for (int k = 0; k < K; ++k) denom += expf(lrow[k] - mx);
            inv = 1.0f / denom;
            best = codebook_softmax_pick(lrow, best, mx, inv);
            float dot = 0.0f;
            for (int k = 0; k < K; ++k) {
                float dk = 0.0f;
#pragma unroll
                for (int f = 0; f < CB_MAX_F; ++f) if (f < F) dk += D[k * F + f] * G[f];
                dot += expf(lrow[k] - mx) * inv * dk;
            }
            for (int k = 0; k < K; ++k) {
                float dk = 0.0f;
#pragma unroll
                for (int f = 0; f < CB_MAX_F; ++f) if (f < F) dk += D[k * F + f] * G[f];
                grow[k] += expf(lrow[k] - mx) * inv * (dk - dot);
            }
        }
        const double scale = (double)((1.0f - inv) + inv);
        long long* sg = s_gdict + ((size_t)(l - lod_begin) * K + best) * F;
#pragma unroll
        for (int f = 0; f < CB_MAX_F; ++f)
            if (f < F && q[f] != 0)
                atomicAdd(reinterpret_cast<unsigned long long*>(sg + f), (unsigned long long)__double2ll_rn((double)q[f] * scale));
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nl * K * F; e += blockDim.x) {
        const long long x = s_gdict[e];
        if (x != 0) atomicAdd(reinterpret_cast<unsigned long long*>(dict_acc + (size_t)lod_begin * K * F + e), (unsigned long long)x);
    }
}

__global__ void __launch_bounds__(256)
codebook_dict_flush_kernel(SgLods ml, int num_lods, int KF, int clog, const SgHeader* __restrict__ hdr,
                           long long* __restrict__ dict_acc) {
    const SgScale sc = sg_scale(hdr->absmax_bits, clog);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < num_lods * KF; e += gridDim.x * blockDim.x) {
        const long long q = dict_acc[e];
        if (q == 0) continue;
        dict_acc[e] = 0;
        ml.grad_dict[e / KF][e % KF] += (float)((double)q * sc.to_float);
    }
}

// ---------------------------------------------------------------------------------------------- host side
struct SgPlan { int64_t total_rows, dict_elems, off_dict, off_flags, off_acc, bytes; int stride; };

static SgPlan sg_plan(int64_t total_rows, int channels, int64_t dict_elems) {
    SgPlan p;
    p.total_rows = total_rows; p.dict_elems = dict_elems; p.stride = sg_stride(channels);
    p.off_dict = (int64_t)sizeof(SgHeader);
    p.off_flags = p.off_dict + sg_round64(dict_elems * 8);
    p.off_acc = p.off_flags + sg_round64(total_rows);
    p.bytes = p.off_acc + total_rows * p.stride * 8;
    return p;
}

extern "C" int64_t wisp_spc_bwd_workspace_bytes(int64_t total_rows, int channels, int64_t dict_elems) {
    if (total_rows < 0 || channels < 1 || dict_elems < 0) return -1;
    return sg_plan(total_rows, channels, dict_elems).bytes;
}

static int sg_clog(int64_t n) {                                           // ceil(log2(8 n)): adds one accumulator can receive
    int c = 3;
    while (((int64_t)1 << (c - 3)) < n) ++c;
    return c;
}

struct SgCall {
    const float* coords; const void* cells; int cells_is_i64; int64_t cell_stride; int spv;
    const int16_t* points; const int32_t* trinkets; const float* grad_out; int64_t n; int num_lods; int channels; int sum;
    int direct_stride;                                                    // row stride of the gradient tensors (non-finite path)
};

// Touched-row flags (one byte store per scattered row, one byte read per table row in pass 3) pay off when a launch reaches a
// small part of a big table - NGLOD's 512 coordinates in a million rows; when most rows are touched anyway (a 2 M-sample NeRF
// batch over 0.2 M rows) pass 3 reads the accumulators themselves and pass 2 saves its 2.5 M scattered byte stores.
static bool sg_use_flags(const SgCall& c, const SgPlan& pl) { return c.n * 8 * c.num_lods < pl.total_rows; }

// few samples (an NGLOD step has 512): the levels of a sample go to different lane groups, or a handful of workgroups would walk
// all levels one after the other with the rest of the chip idle
static int sg_split_lods(const SgCall& c) { return c.num_lods > 1 && c.n * (c.channels <= 64 ? c.channels : 64) < (int64_t)256 * 1024; }

// passes 1 and 2
static int sg_scatter(const SgCall& c, const SgLods& ml, const SgPlan& pl, unsigned char* ws, hipStream_t s) {
    SgHeader* hdr = reinterpret_cast<SgHeader*>(ws);
    uint8_t* flags = sg_use_flags(c, pl) ? ws + pl.off_flags : nullptr;
    long long* acc = reinterpret_cast<long long*>(ws + pl.off_acc);
    const int clog = sg_clog(c.n);
    if (hipMemsetAsync(hdr, 0, sizeof(SgHeader), s) != hipSuccess) return -1;
    const unsigned g1 = (unsigned)min64(ceil_div64(c.n, 256), 4096);
#define SG_ABSMAX(I) hipLaunchKernelGGL((spc_grad_absmax_kernel<I>), dim3(g1), dim3(256), 0, s, c.coords, (const I*)c.cells,   \
                                        c.cell_stride, c.spv, c.points, ml, c.grad_out, c.n, c.num_lods, c.channels, c.sum, hdr)
    if (c.cells_is_i64) SG_ABSMAX(int64_t); else SG_ABSMAX(int32_t);
#undef SG_ABSMAX
    if (c.channels <= 8) {
        const dim3 grid((unsigned)ceil_div64(c.n, SG_THREADS)), block(SG_THREADS);
#define SG_MERGE(FF, I) hipLaunchKernelGGL((spc_grad_scatter_merge_kernel<FF, I>), grid, block, 0, s, c.coords, (const I*)c.cells, \
                                           c.cell_stride, c.spv, c.points, c.trinkets, ml, c.grad_out, c.n, c.num_lods, c.sum, \
                                           clog, pl.stride, c.direct_stride, hdr, flags, acc)
#define SG_CASE(FF) case FF: if (c.cells_is_i64) SG_MERGE(FF, int64_t); else SG_MERGE(FF, int32_t); break;
        switch (c.channels) { SG_CASE(1) SG_CASE(2) SG_CASE(3) SG_CASE(4) SG_CASE(5) SG_CASE(6) SG_CASE(7) SG_CASE(8) }
#undef SG_CASE
#undef SG_MERGE
    } else {
        const int cpt = c.channels <= 64 ? c.channels : 64;
        const int split = sg_split_lods(c);
        const dim3 grid((unsigned)min64(ceil_div64(c.n * (split ? c.num_lods : 1), 256 / cpt), 16384)), block(256);
#define SG_WIDE(I) hipLaunchKernelGGL((spc_grad_scatter_wide_kernel<I>), grid, block, 0, s, c.coords, (const I*)c.cells,         \
                                      c.cell_stride, c.spv, c.points, c.trinkets, ml, c.grad_out, c.n, c.num_lods, c.channels,  \
                                      c.sum, clog, pl.stride, c.direct_stride, hdr, flags, acc, split)
        if (c.cells_is_i64) SG_WIDE(int64_t); else SG_WIDE(int32_t);
#undef SG_WIDE
    }
    return 0;
}

static int sg_fill(SgLods& ml, const int32_t* levels, const int64_t* rows, int num_lods) {
    for (int l = 0; l < SG_MAX_LODS; ++l) {
        ml.grad[l] = nullptr; ml.logits[l] = nullptr; ml.dict[l] = nullptr; ml.grad_dict[l] = nullptr; ml.level[l] = 0; ml.base[l] = 0;
    }
    ml.base[SG_MAX_LODS] = 0;
    int64_t b = 0;
    for (int l = 0; l < num_lods; ++l) {
        if (levels[l] < 0 || levels[l] > 15 || rows[l] < 0) return -1;
        ml.level[l] = levels[l];
        ml.base[l] = b;
        b += rows[l];
    }
    for (int l = num_lods; l <= SG_MAX_LODS; ++l) ml.base[l] = b;
    return 0;
}

static int spc_bwd_impl(const SgCall& c, const int32_t* levels, const int64_t* rows, float* const* grad_feats, void* workspace,
                        int64_t workspace_bytes, hipStream_t s) {
    SgLods ml;
    if (sg_fill(ml, levels, rows, c.num_lods) != 0) return wisp_fail(WISP_ERR_INVALID, __func__, "bad level or row count");
    for (int l = 0; l < c.num_lods; ++l) {
        if (!grad_feats[l]) return wisp_fail(WISP_ERR_INVALID, __func__, "null gradient pointer");
        ml.grad[l] = grad_feats[l];
    }
    const SgPlan pl = sg_plan(ml.base[c.num_lods], c.channels, 0);
    if (!workspace || workspace_bytes < pl.bytes) return wisp_fail(WISP_ERR_INVALID, __func__, "workspace missing or too small (wisp_spc_bwd_workspace_bytes)");
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    if (sg_scatter(c, ml, pl, ws, s) != 0) return wisp_fail(WISP_ERR_LAUNCH, __func__, "hipMemsetAsync failed");
    int lpr = 1;
    while (lpr < c.channels && lpr < 64) lpr <<= 1;
    const unsigned g3 = (unsigned)min64(ceil_div64(pl.total_rows, 256 / lpr), 8192);
    if (pl.total_rows > 0)
        hipLaunchKernelGGL(spc_grad_finalize_kernel, dim3(g3), dim3(256), 0, s, ml, c.num_lods, c.channels, pl.stride, lpr,
                           sg_clog(c.n), reinterpret_cast<const SgHeader*>(ws), sg_use_flags(c, pl) ? ws + pl.off_flags : nullptr,
                           reinterpret_cast<long long*>(ws + pl.off_acc));
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e_));
    return WISP_OK;
}

extern "C" int wisp_spc_trilinear_bwd(const float* coords, const void* pidx, int pidx_is_i64, const int16_t* points,
                                      const int32_t* trinkets, const float* grad_out, int64_t num_voxels,
                                      int samples_per_voxel, int channels, int level, int64_t num_rows, float* grad_feats,
                                      void* workspace, int64_t workspace_bytes, wisp_stream_t stream) {
    WISP_REQUIRE(num_voxels >= 0 && samples_per_voxel >= 1 && channels >= 1 && level >= 0 && level <= 15 && num_rows >= 0, "bad sizes");
    if (num_voxels == 0) return WISP_OK;
    WISP_REQUIRE(coords && pidx && points && trinkets && grad_out && grad_feats, "null pointer");
    WISP_REQUIRE(num_voxels * samples_per_voxel < ((int64_t)1 << 40), "too many samples for one launch");
    SgCall c{coords, pidx, pidx_is_i64, 1, samples_per_voxel, points, trinkets, grad_out, num_voxels * samples_per_voxel, 1,
             channels, 1, channels};
    const int32_t lv[1] = {level};
    const int64_t rw[1] = {num_rows};
    float* const gp[1] = {grad_feats};
    return spc_bwd_impl(c, lv, rw, gp, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int wisp_spc_trilinear_multi_bwd(const float* coords, const int64_t* chain, int64_t chain_stride,
                                            const int16_t* points, const int32_t* trinkets, const float* grad_out,
                                            int64_t num_samples, int num_lods, const int32_t* levels, const int64_t* rows,
                                            int channels, int sum, float* const* grad_feats, void* workspace,
                                            int64_t workspace_bytes, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0 && num_lods >= 1 && num_lods <= SG_MAX_LODS && channels >= 1 && chain_stride >= num_lods, "bad sizes");
    if (num_samples == 0) return WISP_OK;
    WISP_REQUIRE(coords && chain && points && trinkets && grad_out && levels && rows && grad_feats, "null pointer");
    WISP_REQUIRE(num_samples < ((int64_t)1 << 40), "too many samples for one launch");
    SgCall c{coords, chain, 1, chain_stride, 1, points, trinkets, grad_out, num_samples, num_lods, channels, sum, channels};
    return spc_bwd_impl(c, levels, rows, grad_feats, workspace, workspace_bytes, (hipStream_t)stream);
}

static int codebook_bwd_impl(const SgCall& c, const int32_t* levels, const int64_t* rows, const float* const* logits,
                             const float* const* dictionaries, int dict_size, float* const* grad_logits,
                             float* const* grad_dictionaries, void* workspace, int64_t workspace_bytes, hipStream_t s) {
    const int K = dict_size, F = c.channels;
    SgLods ml;
    if (sg_fill(ml, levels, rows, c.num_lods) != 0) return wisp_fail(WISP_ERR_INVALID, __func__, "bad level or row count");
    for (int l = 0; l < c.num_lods; ++l) {
        if (!logits[l] || !dictionaries[l] || !grad_logits[l] || !grad_dictionaries[l])
            return wisp_fail(WISP_ERR_INVALID, __func__, "null table pointer");
        ml.grad[l] = grad_logits[l]; ml.logits[l] = logits[l]; ml.dict[l] = dictionaries[l]; ml.grad_dict[l] = grad_dictionaries[l];
    }
    const int64_t dict_elems = (int64_t)c.num_lods * K * F;
    const SgPlan pl = sg_plan(ml.base[c.num_lods], F, dict_elems);
    if (!workspace || workspace_bytes < pl.bytes) return wisp_fail(WISP_ERR_INVALID, __func__, "workspace missing or too small (wisp_spc_bwd_workspace_bytes)");
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    // (non-finite path: G cannot be scattered into the logits gradient rows - they are K wide - so it goes to the first F
    //  columns, which is where a non-finite value has to show up for the found-inf check; see the header)
    if (sg_scatter(c, ml, pl, ws, s) != 0) return wisp_fail(WISP_ERR_LAUNCH, __func__, "hipMemsetAsync failed");
    const SgHeader* hdr = reinterpret_cast<const SgHeader*>(ws);
    uint8_t* flags = sg_use_flags(c, pl) ? ws + pl.off_flags : nullptr;
    long long* acc = reinterpret_cast<long long*>(ws + pl.off_acc);
    long long* dict_acc = reinterpret_cast<long long*>(ws + pl.off_dict);
    const int clog = sg_clog(c.n);
    // all levels in one launch while their dictionaries (+ fixed-point gradients) fit the default LDS window, else level groups
    const size_t per_level = (size_t)K * F * 12;
    int group = (int)(48 * 1024 / per_level);
    if (group < 1) group = 1;
    bool rows16 = true;                                                   // 16-byte loads of the logits rows need aligned tensors
    for (int l = 0; l < c.num_lods; ++l)
        rows16 = rows16 && ((reinterpret_cast<uintptr_t>(logits[l]) | reinterpret_cast<uintptr_t>(grad_logits[l])) & 15) == 0;
    for (int lb = 0; lb < c.num_lods; lb += group) {
        const int le = lb + group < c.num_lods ? lb + group : c.num_lods;
        const int64_t nrows = ml.base[le] - ml.base[lb];
        if (nrows == 0) continue;
        const unsigned g3 = (unsigned)min64(ceil_div64(nrows, 256), 2048);
        if (K % 4 == 0 && K <= 16 && rows16)
            hipLaunchKernelGGL(codebook_grad_finalize_kernel<16>, dim3(g3), dim3(256), (size_t)(le - lb) * per_level, s, ml, lb, le, K, F,
                               pl.stride, clog, hdr, flags, acc, dict_acc);
        else
            hipLaunchKernelGGL(codebook_grad_finalize_kernel<0>, dim3(g3), dim3(256), (size_t)(le - lb) * per_level, s, ml, lb, le, K, F,
                               pl.stride, clog, hdr, flags, acc, dict_acc);
    }
    hipLaunchKernelGGL(codebook_dict_flush_kernel, dim3((unsigned)min64(ceil_div64(dict_elems, 256), 64)), dim3(256), 0, s, ml,
                       c.num_lods, K * F, clog, hdr, dict_acc);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e_));
    return WISP_OK;
}

extern "C" int wisp_codebook_trilinear_bwd(const float* coords, const void* pidx, int pidx_is_i64, const int16_t* points,
                                           const int32_t* trinkets, const float* logits, const float* dictionary,
                                           const float* grad_out, int64_t num_voxels, int samples_per_voxel, int dict_size,
                                           int feature_dim, int level, int64_t num_logit_rows, float* grad_logits,
                                           float* grad_dictionary, void* workspace, int64_t workspace_bytes,
                                           wisp_stream_t stream) {
    WISP_REQUIRE(num_voxels >= 0 && samples_per_voxel >= 1 && level >= 0 && level <= 15 && num_logit_rows >= 0, "bad sizes");
    WISP_REQUIRE(dict_size >= 1 && dict_size <= CB_MAX_K && feature_dim >= 1 && feature_dim <= CB_MAX_F && feature_dim <= dict_size,
                 "dictionary shape outside the fused kernels (dict_size <= 256, feature_dim <= min(16, dict_size))");
    if (num_voxels == 0) return WISP_OK;
    WISP_REQUIRE(coords && pidx && points && trinkets && logits && dictionary && grad_out && grad_logits && grad_dictionary, "null pointer");
    SgCall c{coords, pidx, pidx_is_i64, 1, samples_per_voxel, points, trinkets, grad_out, num_voxels * samples_per_voxel, 1,
             feature_dim, 1, dict_size};
    const int32_t lv[1] = {level};
    const int64_t rw[1] = {num_logit_rows};
    const float* const lg[1] = {logits};
    const float* const dc[1] = {dictionary};
    float* const gl[1] = {grad_logits};
    float* const gd[1] = {grad_dictionary};
    return codebook_bwd_impl(c, lv, rw, lg, dc, dict_size, gl, gd, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int wisp_codebook_trilinear_multi_bwd(const float* coords, const int64_t* chain, int64_t chain_stride,
                                                 const int16_t* points, const int32_t* trinkets, const float* const* logits,
                                                 const float* const* dictionaries, const float* grad_out, int64_t num_samples,
                                                 int num_lods, const int32_t* levels, const int64_t* rows, int dict_size,
                                                 int feature_dim, int sum, float* const* grad_logits,
                                                 float* const* grad_dictionaries, void* workspace, int64_t workspace_bytes,
                                                 wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0 && num_lods >= 1 && num_lods <= SG_MAX_LODS && chain_stride >= num_lods, "bad sizes");
    WISP_REQUIRE(dict_size >= 1 && dict_size <= CB_MAX_K && feature_dim >= 1 && feature_dim <= CB_MAX_F && feature_dim <= dict_size,
                 "dictionary shape outside the fused kernels (dict_size <= 256, feature_dim <= min(16, dict_size))");
    if (num_samples == 0) return WISP_OK;
    WISP_REQUIRE(coords && chain && points && trinkets && logits && dictionaries && grad_out && levels && rows && grad_logits
                 && grad_dictionaries, "null pointer");
    SgCall c{coords, chain, 1, chain_stride, 1, points, trinkets, grad_out, num_samples, num_lods, feature_dim, sum, dict_size};
    return codebook_bwd_impl(c, levels, rows, logits, dictionaries, dict_size, grad_logits, grad_dictionaries, workspace,
                             workspace_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- fused SDF regression step
// One optimisation step of the reference's SDFTrainer (wisp/trainers/sdf_trainer.py:65-124, only_last) for a NeuralSDF over an
// OctreeGrid (wisp/models/nefs/neural_sdf.py:102-155; nglod_octree.yaml: 16 'sum' features on 6 levels, [position, features]
// -> Linear(19, 128) -> relu -> Linear(128, 1), 512 coordinates per step):
//     pred = decoder([x, sum_l trilinear_l(x)]) ;  loss = sum (pred - gt)^2 / B ;  backward
// The reference issues ~60 kernels for it (a query, six trilinear lookups, two GEMMs with their elementwise companions, the
// loss, and all of that again backwards); the modular path of this package 20.  At 512 coordinates every one of them is
// launch latency.  Here:
//   sdf_train_kernel     16 lanes per (sample, level) walk the octree to the level's cell and blend its corners (the statements
//                        of sdf_trace_fused_kernel: bit-identical features; all levels of a sample at once, so the dependent
//                        loads of the walk cost the deepest level's, not the sum); then 16 lanes per sample run the decoder
//                        forward, d loss / d pred and the decoder backward: lane c keeps hidden units c, c + 16, ..., the
//                        gradient of the decoder input is a column sum over LDS.  Per workgroup the weight gradients are
//                        summed by the thread that owns the entry, in sample order, and stored as one partial row - no atomics.  The
//                        sample's feature gradient [16] and cell chain go to scratch; the workgroup's largest |w g| goes to
//                        the header of the order-free scatter below, which therefore needs no magnitude pass of its own.
//   sdf_train_reduce     adds the partial rows up in workgroup order, ADDS the sums to the decoder's gradient tensors, writes
//                        the loss.
//   spc_grad_scatter_wide + spc_grad_finalize   the corner sums of all six levels (64-bit fixed point, above).
// Every sum has a fixed order or is an integer sum: the step is bitwise repeatable.
#define ST_GROUP 16
#define ST_GROUPS 16                       // samples per workgroup pass
#define ST_MAX_HIDDEN 256
#define ST_MAX_IN 32

struct StField {
    const float* feats[SG_MAX_LODS];
    int32_t level[SG_MAX_LODS];
    int num_lods, channels, half_round, hidden, max_level;
    const float *w1, *b1, *w2, *b2;
};

static __device__ __forceinline__ int st_child_slot(int qx, int qy, int qz, int sh) {
    return (((qx >> sh) & 1) << 2) | (((qy >> sh) & 1) << 1) | ((qz >> sh) & 1);
}

__global__ void __launch_bounds__(ST_GROUP * ST_GROUPS)
sdf_train_kernel(const float* __restrict__ coords, const float* __restrict__ gts, int64_t n, const uint8_t* __restrict__ octree,
                 const int32_t* __restrict__ exsum, const int16_t* __restrict__ points, const int32_t* __restrict__ trinkets,
                 StField fld, float inv_batch, float* __restrict__ partials /* [grid][row] */, int row_stride,
                 float* __restrict__ dfeat /* [n][channels] */, int64_t* __restrict__ chain /* [n][num_lods] */,
                 SgHeader* __restrict__ hdr) {
    extern __shared__ float s_st[];
    const int C = fld.channels, H = fld.hidden, NL = fld.num_lods;
    const int in_dim = 3 + C;
    const int in_pad = in_dim | 1;                      // odd row stride: the lanes of a group read different rows
    const int spb = ST_GROUPS / NL;                     // samples per workgroup pass: one lane group per (sample, level)
    float* s_w1 = s_st;                                 // [H][in_pad]
    float* s_b1 = s_w1 + H * in_pad;                    // [H]
    float* s_w2 = s_b1 + H;                             // [H]
    float* s_in = s_w2 + H;                             // [spb][in_dim]        decoder inputs of the pass
    float* s_ga = s_in + ST_GROUPS * in_dim;            // [spb][H]             d loss / d pre-activation
    float* s_gr = s_ga + ST_GROUPS * H;                 // [spb][H]             d loss / d pred * relu output (for d w2)
    float* s_g = s_gr + ST_GROUPS * H;                  // [spb]                d loss / d pred
    float* s_sq = s_g + ST_GROUPS;                      // [spb]                squared error
    float* s_lev = s_sq + ST_GROUPS;                    // [groups][C]          one level's lookup of one sample
    float* s_wm = s_lev + ST_GROUPS * ST_GROUP;         // [groups]             its largest corner weight
    for (int e = threadIdx.x; e < H * in_dim; e += blockDim.x) s_w1[(e / in_dim) * in_pad + e % in_dim] = fld.w1[e];
    for (int e = threadIdx.x; e < H; e += blockDim.x) { s_b1[e] = fld.b1[e]; s_w2[e] = fld.w2[e]; }
    const int c = threadIdx.x & (ST_GROUP - 1);
    const int grp = threadIdx.x / ST_GROUP;
    const int L = fld.max_level;
    const float b2 = fld.b2[0];
    // partial sums of this workgroup: d W1 [H][in_dim], d b1 [H], d w2 [H], d b2, loss - in that order; every entry has one owner
    const int n_entries = H * in_dim + 2 * H + 2;
    float* s_own = s_wm + ST_GROUPS;                    // [n_entries]
    for (int e = threadIdx.x; e < n_entries; e += blockDim.x) s_own[e] = 0.0f;
    uint32_t mbits = 0;
    const int my_si = grp / NL, my_li = grp - my_si * NL;                 // phase 1: this group's (sample of the pass, level)
    __syncthreads();
    for (int64_t base = (int64_t)blockIdx.x * spb; base < n; base += (int64_t)gridDim.x * spb) {
        // ---- phase 1: every (sample, level) pair on its own lane group: walk to the level's cell, trilinear lookup.  The walk
        //      is a chain of dependent loads; six groups walking at once cost the time of the deepest one
        {
            const int64_t s = base + my_si;
            float acc = 0.0f, wm = 0.0f;
            if (my_si < spb && s < n) {
                const float px = coords[s * 3], py = coords[s * 3 + 1], pz = coords[s * 3 + 2];
                const bool inside = (fabsf(px) <= 1.0f) && (fabsf(py) <= 1.0f) && (fabsf(pz) <= 1.0f);
                const float res = (float)(1 << L);
                const int top = (1 << L) - 1;
                const int qx = min((int)floorf(res * (0.5f * px + 0.5f)), top);
                const int qy = min((int)floorf(res * (0.5f * py + 0.5f)), top);
                const int qz = min((int)floorf(res * (0.5f * pz + 0.5f)), top);
                const float pos[3] = {px, py, pz};
                const int lv = fld.level[my_li];
                int64_t node = inside ? 0 : -1;
                for (int l = 0; l < lv && node >= 0; ++l) {               // (spc_query_kernel's walk)
                    const int cs = st_child_slot(qx, qy, qz, L - 1 - l);
                    const uint32_t bits = octree[node];                   // both loads of a level go out together: one round
                    const int64_t first = (int64_t)exsum[node];           // trip per level, not two
                    const int64_t child = first + __popc(bits & ((2u << cs) - 1u));
                    node = ((bits >> cs) & 1u) ? child : -1;
                }
                if (c == 0) chain[s * NL + my_li] = node;
                if (node >= 0) {
                    float w[8];
                    sg_coeffs(pos, points + node * 3, lv, w);
#pragma unroll
                    for (int j = 0; j < 8; ++j) wm = fmaxf(wm, fabsf(w[j]));
                    const int32_t* tr = trinkets + node * 8;
                    const float* f = fld.feats[my_li];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float fv = f[(int64_t)tr[j] * C + c];
                        if (fld.half_round) fv = __half2float(__float2half_rn(fv));
                        acc += fv * w[j];
                    }
                    if (fld.half_round) acc = __half2float(__float2half_rn(acc));
                }
            }
            s_lev[grp * ST_GROUP + c] = acc;
            if (c == 0) s_wm[grp] = wm;
        }
        __syncthreads();
        // ---- phase 2: group si owns sample si of the pass: decoder forward, d loss / d pred, decoder backward
        const int64_t s = base + grp;
        const bool live = grp < spb && s < n;
        float* gin = s_in + grp * in_dim;
        float* ga = s_ga + grp * H;
        float* gr = s_gr + grp * H;
        float wmax = 0.0f;
        if (live) {
            float feat = 0.0f;                                            // channel c, summed over the levels in order
            for (int li = 0; li < NL; ++li) {
                feat += s_lev[(grp * NL + li) * ST_GROUP + c];
                wmax = fmaxf(wmax, s_wm[grp * NL + li]);
            }
            if (c < 3) gin[c] = coords[s * 3 + c];
            gin[3 + c] = feat;
        }
        __builtin_amdgcn_wave_barrier();                 // a group's lanes are in one wave: LDS order suffices
        float g = 0.0f;
        if (live) {
            // decoder forward: in = [position, features] (neural_sdf.py: embedded position first)
            float out = 0.0f;
            for (int hh = c; hh < H; hh += ST_GROUP) {
                const float* wr = s_w1 + hh * in_pad;
                float a = s_b1[hh];
                for (int i = 0; i < in_dim; ++i) a = __builtin_fmaf(wr[i], gin[i], a);
                const float r = fmaxf(a, 0.0f);
                out = __builtin_fmaf(s_w2[hh], r, out);
                ga[hh] = a > 0.0f ? s_w2[hh] : 0.0f;    // finished below, once d loss / d pred is known
                gr[hh] = r;
            }
#pragma unroll
            for (int d = ST_GROUP / 2; d >= 1; d >>= 1) out += __shfl_xor(out, d, ST_GROUP);
            const float diff = (out + b2) - gts[s];
            g = 2.0f * diff * inv_batch;                 // d [sum (pred - gt)^2 / B] / d pred
            for (int hh = c; hh < H; hh += ST_GROUP) { ga[hh] *= g; gr[hh] *= g; }
            if (c == 0) { s_g[grp] = g; s_sq[grp] = diff * diff; }
        } else if (grp < spb) {
            for (int hh = c; hh < H; hh += ST_GROUP) { ga[hh] = 0.0f; gr[hh] = 0.0f; }
            if (c < 3) gin[c] = 0.0f;
            gin[3 + c] = 0.0f;
            if (c == 0) { s_g[grp] = 0.0f; s_sq[grp] = 0.0f; }
        }
        __builtin_amdgcn_wave_barrier();
        if (live) {
            // gradient of the decoder input, feature columns only (nothing consumes d / d position)
            float dx = 0.0f;
            for (int hh = 0; hh < H; ++hh) dx = __builtin_fmaf(ga[hh], s_w1[hh * in_pad + 3 + c], dx);
            dfeat[s * C + c] = dx;
            const uint32_t b = __float_as_uint(wmax * fabsf(dx)) & 0x7fffffffu;
            const uint32_t nb = (!(dx == dx) || !(wmax == wmax)) ? 0x7fc00000u : b;
            mbits = nb > mbits ? nb : mbits;
        }
        __syncthreads();
        // ---- phase 3: weight gradients of this pass: the thread that owns an entry adds the samples up in order
        for (int e = threadIdx.x; e < n_entries; e += blockDim.x) {
            float acc = 0.0f;
            if (e < H * in_dim) {
                const int hh = e / in_dim, i = e - hh * in_dim;
                for (int q = 0; q < spb; ++q) acc = __builtin_fmaf(s_ga[q * H + hh], s_in[q * in_dim + i], acc);
            } else if (e < H * in_dim + H) {
                const int hh = e - H * in_dim;
                for (int q = 0; q < spb; ++q) acc += s_ga[q * H + hh];
            } else if (e < H * in_dim + 2 * H) {
                const int hh = e - H * in_dim - H;
                for (int q = 0; q < spb; ++q) acc += s_gr[q * H + hh];
            } else if (e == H * in_dim + 2 * H) {
                for (int q = 0; q < spb; ++q) acc += s_g[q];
            } else {
                for (int q = 0; q < spb; ++q) acc += s_sq[q];
            }
            s_own[e] += acc;
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < n_entries; e += blockDim.x) partials[(int64_t)blockIdx.x * row_stride + e] = s_own[e];
    mbits = sg_wave_umax(mbits);
    if ((threadIdx.x & 63) == 0 && mbits > __atomic_load_n(&hdr->absmax_bits, __ATOMIC_RELAXED)) atomicMax(&hdr->absmax_bits, mbits);
}

__global__ void __launch_bounds__(256)
sdf_train_reduce_kernel(const float* __restrict__ partials, int rows, int row_stride, int H, int in_dim, float* __restrict__ gw1,
                        float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2, float* __restrict__ loss,
                        float inv_batch) {
    // 16 lanes per entry: lane r adds rows r, r + 16, ... in order, then a fixed butterfly over the 16 lanes
    const int n_entries = H * in_dim + 2 * H + 2;
    const int r0 = threadIdx.x & 15;
    for (int e = blockIdx.x * 16 + (threadIdx.x >> 4); e < n_entries + 15; e += gridDim.x * 16) {      // (whole groups stay together)
        const bool in = e < n_entries;
        float acc = 0.0f;
        if (in)
            for (int r = r0; r < rows; r += 16) acc += partials[(int64_t)r * row_stride + e];
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 16);
        if (!in || r0 != 0) continue;
        if (e < H * in_dim) gw1[e] += acc;
        else if (e < H * in_dim + H) gb1[e - H * in_dim] += acc;
        else if (e < H * in_dim + 2 * H) gw2[e - H * in_dim - H] += acc;
        else if (e == H * in_dim + 2 * H) gb2[0] += acc;
        else loss[0] = acc * inv_batch;
    }
}

static inline int st_grid(int64_t n, int num_lods) { return (int)min64(ceil_div64(n, ST_GROUPS / num_lods), 1024); }
static inline int st_row_stride(int hidden, int channels) { return (hidden * (3 + channels) + 2 * hidden + 2 + 15) / 16 * 16; }

extern "C" int64_t wisp_sdf_train_scratch_bytes(int64_t n, int num_lods, int channels, int hidden) {
    if (n < 0 || num_lods < 1 || num_lods > SG_MAX_LODS || channels < 1 || hidden < 1) return -1;
    return (int64_t)st_grid(n, num_lods) * st_row_stride(hidden, channels) * 4 + sg_round64(n * channels * 4) + n * num_lods * 8;
}

extern "C" int wisp_sdf_train_step(const float* coords, const float* gts, int64_t n, const uint8_t* octree, const int32_t* exsum,
                                   const int16_t* points, const int32_t* trinkets, const float* const* feats,
                                   const int32_t* levels, const int64_t* rows, int num_lods, int channels, int half_round,
                                   const float* w1, const float* b1, const float* w2, const float* b2, int hidden,
                                   float* const* grad_feats, float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2,
                                   float* loss, void* scratch, int64_t scratch_bytes, void* workspace, int64_t workspace_bytes,
                                   wisp_stream_t stream) {
    WISP_REQUIRE(n >= 1 && num_lods >= 1 && num_lods <= SG_MAX_LODS, "bad sizes");
    WISP_REQUIRE(channels == ST_GROUP, "the fused SDF step is built for 16 feature channels (nglod_octree.yaml)");
    WISP_REQUIRE(hidden >= 1 && hidden <= ST_MAX_HIDDEN, "hidden width out of range");
    WISP_REQUIRE(coords && gts && octree && exsum && points && trinkets && feats && levels && rows && w1 && b1 && w2 && b2 &&
                 grad_feats && grad_w1 && grad_b1 && grad_w2 && grad_b2 && loss && scratch && workspace, "null pointer");
    WISP_REQUIRE(scratch_bytes >= wisp_sdf_train_scratch_bytes(n, num_lods, channels, hidden), "scratch too small (wisp_sdf_train_scratch_bytes)");
    StField fld;
    SgLods ml;
    WISP_REQUIRE(sg_fill(ml, levels, rows, num_lods) == 0, "bad level or row count");
    for (int l = 0; l < num_lods; ++l) {
        WISP_REQUIRE(feats[l] && grad_feats[l] && (l == 0 || levels[l] > levels[l - 1]), "bad level list");
        fld.feats[l] = feats[l]; fld.level[l] = levels[l]; ml.grad[l] = grad_feats[l];
    }
    fld.num_lods = num_lods; fld.channels = channels; fld.half_round = half_round; fld.hidden = hidden;
    fld.max_level = levels[num_lods - 1];
    fld.w1 = w1; fld.b1 = b1; fld.w2 = w2; fld.b2 = b2;
    const SgPlan pl = sg_plan(ml.base[num_lods], channels, 0);
    WISP_REQUIRE(workspace_bytes >= pl.bytes, "workspace too small (wisp_spc_bwd_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    SgHeader* hdr = reinterpret_cast<SgHeader*>(ws);
    const int grid = st_grid(n, num_lods), row_stride = st_row_stride(hidden, channels), in_dim = 3 + channels;
    float* partials = static_cast<float*>(scratch);
    float* dfeat = partials + (size_t)grid * row_stride;
    int64_t* chain = reinterpret_cast<int64_t*>(reinterpret_cast<unsigned char*>(dfeat) + sg_round64(n * channels * 4));
    const size_t lds = ((size_t)hidden * (in_dim | 1) + 2 * hidden + (size_t)ST_GROUPS * (in_dim + 2 * hidden + 2 + ST_GROUP + 1) +
                        (size_t)hidden * in_dim + 2 * hidden + 2) * 4;
    const float inv_batch = 1.0f / (float)n;
    if (hipMemsetAsync(hdr, 0, sizeof(SgHeader), s) != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, __func__, "hipMemsetAsync failed");
    if (const hipError_t e = WISP_ALLOW_LDS(sdf_train_kernel, lds)) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
    hipLaunchKernelGGL(sdf_train_kernel, dim3(grid), dim3(ST_GROUP * ST_GROUPS), lds, s, coords, gts, n, octree, exsum, points,
                       trinkets, fld, inv_batch, partials, row_stride, dfeat, chain, hdr);
    hipLaunchKernelGGL(sdf_train_reduce_kernel, dim3((hidden * in_dim + 2 * hidden + 2 + 15) / 16), dim3(256), 0, s,
                       partials, grid, row_stride, hidden, in_dim, grad_w1, grad_b1, grad_w2, grad_b2, loss, inv_batch);
    // the corner sums: the magnitude bound is in the header already (sdf_train_kernel), so only scatter + row pass
    SgCall c{coords, chain, 1, num_lods, 1, points, trinkets, dfeat, n, num_lods, channels, 1, channels};
    uint8_t* flags = sg_use_flags(c, pl) ? ws + pl.off_flags : nullptr;
    long long* acc = reinterpret_cast<long long*>(ws + pl.off_acc);
    const int clog = sg_clog(n);
    const int split = sg_split_lods(c);
    hipLaunchKernelGGL((spc_grad_scatter_wide_kernel<int64_t>),
                       dim3((unsigned)min64(ceil_div64(n * (split ? num_lods : 1), 256 / channels), 16384)), dim3(256), 0, s, coords, chain,
                       (int64_t)num_lods, 1, points, trinkets, ml, dfeat, n, num_lods, channels, 1, clog, pl.stride, channels, hdr, flags,
                       acc, split);
    int lpr = 1;
    while (lpr < channels && lpr < 64) lpr <<= 1;
    if (pl.total_rows > 0)
        hipLaunchKernelGGL(spc_grad_finalize_kernel, dim3((unsigned)min64(ceil_div64(pl.total_rows, 256 / lpr), 8192)), dim3(256), 0, s, ml,
                           num_lods, channels, pl.stride, lpr, clog, hdr, flags, acc);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}
