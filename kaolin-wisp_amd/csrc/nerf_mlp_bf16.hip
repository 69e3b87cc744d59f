// bf16 matrix-core path of the fused radiance-field decoder (see nerf_mlp.hip for what it replaces in the reference).
//
// Design: activations never leave registers between layers.
//   * One wave = one tile of 32 samples; sample n = lane & 31 sits on the N side of v_mfma_f32_32x32x16_bf16, the layer's
//     weights are the A operand (LDS, rows padded for conflict-free ds_read_b128), so after a layer lane (n, g = lane >> 5)
//     holds 16 output neurons of ITS sample per 32-row block: accumulator register r <-> neuron 8 (r / 4) + 4 g + r % 4.
//   * The B operand of the next layer wants 8 K-values per lane.  The order of K inside a contraction is free as long as
//     A and B agree, so accumulator registers 8 h .. 8 h + 7 of block t ARE K-block 2 t + h of the next layer after a
//     v_cvt_pk_bf16_f32; the weights are stored in LDS with their columns permuted to match
//     (slot 16 kb + 8 g + j  <->  neuron phi = 16 kb + 8 (j / 4) + 4 g + j % 4).  No LDS round trip, no shuffles.
//   * relu is one v_pk_max_i16 per two values on the packed bf16; hidden biases enter as the MFMA C operand (free); the
//     bias of the first colour layer rides on a constant-one slot of the view-encoding block.
//   * backward (two wave roles per workgroup, see the comment at mlp_bwd_kernel): the chain wave recomputes the forward,
//     keeps the packed activations in registers for the relu masks and back-propagates through the transposed (also
//     column-permuted) weights with the same chaining; dW = dY^T X is formed by its partner wave with
//     v_mfma_f32_16x16x32_bf16 whose K dimension is the tile's 32 samples.  For that both operands have to be
//     transposed (samples from lanes to registers): the packed registers are written to an LDS image made of
//     8-byte chunks (sample, 4 features) and read back with ds_read_b64_tr_b16.  Chunk address =
//     (fq >> 1) * 640 + (2 n + (fq & 1)) * 8 with fq = feature / 4: both the writes and the transposing reads are
//     bank-conflict free.  dW accumulates in 180 VGPRs for the whole launch; one partial row per workgroup is reduced
//     afterwards.
//   * bias gradients: one extra MFMA per 16 neurons whose A operand is all ones in ONE row - all of them land in one
//     shared 16x16 accumulator block (db3 comes with dW3's ones column).
#include "wisp_common.h"
#include "nerf_mlp_shape.h"
#include <cstdlib>

#include "nerf_mlp_bf16_dev.h"

namespace {
using namespace wisp_mlp;
using namespace wisp_mlp_dev;


// forward operands  [out row][K slots], row stride = K + 8 elements
constexpr int LD1 = 40, LD2 = 72, LD3 = 56, LD4 = 72, LD5 = 72;
constexpr int L_W1 = 0;                     // [64][32]  natural K (the grid features come straight from HBM)
constexpr int L_W2 = L_W1 + 64 * LD1;       // [16][64]  chained K
constexpr int L_W3 = L_W2 + 16 * LD2;       // [64][48]  block 0 chained (density-MLP outputs), blocks 1-2 view encoding
constexpr int L_W4 = L_W3 + 64 * LD3;       // [64][64]  chained
constexpr int L_W5 = L_W4 + 64 * LD4;       // [ 4][64]  chained (3 real rows)
constexpr int L_FWD_END = L_W5 + 4 * LD5;
// backward operands [in row][out-neuron slots]
constexpr int LT5 = 24, LT4 = 72, LT3 = 72, LT2 = 24, LT1 = 72;
constexpr int L_W5T = L_FWD_END;            // [64][16]  slot p < 3 <-> colour channel p
constexpr int L_W4T = L_W5T + 64 * LT5;     // [64][64]
constexpr int L_W3T = L_W4T + 64 * LT4;     // [16][64]  row m <-> density-MLP output m (row 0 unused)
constexpr int L_W2T = L_W3T + 16 * LT3;     // [64][16]
constexpr int L_W1T = L_W2T + 64 * LT2;     // [32][64]
constexpr int L_BWD_END = L_W1T + 32 * LT1;

constexpr int TILE_BYTES = 8 * TILE_REGION; // 64 features x 32 samples

// Prologue: the packed fp32 parameters are first copied to an LDS staging area with coalesced loads (one global
// round trip), then scattered into the permuted bf16 operand images from there.
template <int THREADS>
DEV void stage_params(float* stg, const float* __restrict__ P, int tid, int in_dim) {
    constexpr int PER = (NPARAM + THREADS - 1) / THREADS;
    float v[PER];                                    // all loads in flight before the first LDS write
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int e = tid + k * THREADS; v[k] = e < NPARAM ? packed_param(P, e, in_dim) : 0.0f; }
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int e = tid + k * THREADS; if (e < NPARAM) stg[e] = v[k]; }
}
template <bool BWD>
DEV void stage_weights(__bf16* sw, const float* P, int tid, int nthreads) {
    _Pragma("unroll 4") for (int e = tid; e < 64 * 32; e += nthreads) { const int r = e >> 5, c = e & 31; sw[L_W1 + r * LD1 + c] = (__bf16)P[OW1 + r * IN + c]; }
    _Pragma("unroll 4") for (int e = tid; e < 16 * 64; e += nthreads) { const int r = e >> 6, s = e & 63; sw[L_W2 + r * LD2 + s] = (__bf16)P[OW2 + r * H + phi(s)]; }
    _Pragma("unroll 4") for (int e = tid; e < 64 * 48; e += nthreads) {
        const int r = e / 48, s = e % 48;
        float v = 0.0f;
        if (s < 16) { const int m = phi16(s); if (m) v = P[OW3 + r * X2 + m - 1]; }
        else if (s < ONES_SLOT) v = P[OW3 + r * X2 + s - 1];          // encoding element s - 16 <-> column 15 + (s - 16)
        else if (s == ONES_SLOT) v = P[OB3 + r];
        sw[L_W3 + r * LD3 + s] = (__bf16)v;
    }
    _Pragma("unroll 4") for (int e = tid; e < 64 * 64; e += nthreads) { const int r = e >> 6, s = e & 63; sw[L_W4 + r * LD4 + s] = (__bf16)P[OW4 + r * H + phi(s)]; }
    _Pragma("unroll 4") for (int e = tid; e < 4 * 64; e += nthreads) { const int r = e >> 6, s = e & 63; sw[L_W5 + r * LD5 + s] = (__bf16)(r < 3 ? P[OW5 + r * H + phi(s)] : 0.0f); }
    if (BWD) {
        _Pragma("unroll 4") for (int e = tid; e < 64 * 16; e += nthreads) { const int k = e >> 4, p = e & 15; sw[L_W5T + k * LT5 + p] = (__bf16)(p < 3 ? P[OW5 + p * H + k] : 0.0f); }
        _Pragma("unroll 4") for (int e = tid; e < 64 * 64; e += nthreads) { const int k = e >> 6, s = e & 63; sw[L_W4T + k * LT4 + s] = (__bf16)P[OW4 + phi(s) * H + k]; }
        _Pragma("unroll 4") for (int e = tid; e < 16 * 64; e += nthreads) { const int m = e >> 6, s = e & 63; sw[L_W3T + m * LT3 + s] = (__bf16)(m ? P[OW3 + phi(s) * X2 + m - 1] : 0.0f); }
        _Pragma("unroll 4") for (int e = tid; e < 64 * 16; e += nthreads) { const int k = e >> 4, p = e & 15; sw[L_W2T + k * LT2 + p] = (__bf16)P[OW2 + phi16(p) * H + k]; }
        _Pragma("unroll 4") for (int e = tid; e < 32 * 64; e += nthreads) { const int k = e >> 6, s = e & 63; sw[L_W1T + k * LT1 + s] = (__bf16)P[OW1 + phi(s) * IN + k]; }
    }
}


// everything of one tile that the backward pass needs again
struct Acts {
    bf16x8 x0[2], h1[4], x2[3], h2[4], h3[4];
    float y0, sg[3];
};

struct LaneConst {
    int n, g;
    const __bf16 *w1, *w2, *w3, *w4, *w5;          // this lane's row of each forward operand (+ 8 g)
    floatx16 b1[2], b4[2];                          // hidden biases in accumulator layout (register-resident variant)
    const float* bias_lds;                          // ... or their LDS copy [layer][t][g][16] (+ 16 g)
    float b2[8], b5[3];
};

constexpr int BIASV_FLOATS = 2 * 2 * 2 * 16;
DEV void stage_bias_vectors(float* biasv, const float* P, int tid, int nthreads) {
    for (int e = tid; e < BIASV_FLOATS; e += nthreads) {
        const int r = e & 15, g = (e >> 4) & 1, t = (e >> 5) & 1, layer = e >> 6;
        biasv[e] = P[(layer ? OB4 : OB1) + 32 * t + acc_row(r, g)];
    }
}

template <bool BIAS_LDS>
DEV void lane_const(LaneConst& L, const __bf16* sw, const float* P, int lane, const float* biasv = nullptr) {
    L.n = lane & 31; L.g = lane >> 5;
    L.w1 = sw + L_W1 + L.n * LD1 + 8 * L.g;
    L.w2 = sw + L_W2 + (L.n & 15) * LD2 + 8 * L.g;
    L.w3 = sw + L_W3 + L.n * LD3 + 8 * L.g;
    L.w4 = sw + L_W4 + L.n * LD4 + 8 * L.g;
    L.w5 = sw + L_W5 + (L.n < 3 ? L.n : 3) * LD5 + 8 * L.g;
    L.bias_lds = BIAS_LDS ? biasv + 16 * L.g : nullptr;
    if (!BIAS_LDS) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                L.b1[t][r] = P[OB1 + 32 * t + acc_row(r, L.g)];
                L.b4[t][r] = P[OB4 + 32 * t + acc_row(r, L.g)];
            }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) L.b2[r] = P[OB2 + acc_row(r, L.g)];
#pragma unroll
    for (int c = 0; c < 3; ++c) L.b5[c] = P[OB5 + c];
}

// forward of one tile; x0 and the view direction are already in registers
// PIN: a compiler-level memory barrier in front of every layer.  The operand images never change after staging, so without
// it the optimiser hoists all 26 operand reads (and the bias vectors) out of the tile loop: 256 registers, two waves per SIMD.
// That is what the backward's chain waves want (they share the SIMD with a 180-register accumulator wave anyway); the forward
// kernel is bound by the latency of its input loads and wants waves instead: ~110 registers, four per SIMD.
// CODED: the caller has put the view-direction blocks into A.x2[1..2] already (per-RAY code table, see dir_code_kernel).
template <bool BIAS_LDS, bool PIN = false, bool CODED = false>
DEV void forward_tile(const LaneConst& L, const float d[3], Acts& A) {
    // L1: h1 = relu(W1 x0 + b1)
    if (PIN) asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        floatx16 acc = BIAS_LDS ? *reinterpret_cast<const floatx16*>(L.bias_lds + t * 32) : L.b1[t];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc = mma32(lds_a(L.w1, t * 32 * LD1 + 16 * kb), A.x0[kb], acc);
        A.h1[2 * t] = pack8<0, true>(acc);
        A.h1[2 * t + 1] = pack8<8, true>(acc);
    }
    // L2: y = W2 h1 + b2 (16 rows = accumulator registers 0..7); density = relu(y0); y[1..15] feed the colour MLP
    if (PIN) asm volatile("" ::: "memory");
    {
        floatx16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc = mma32(lds_a(L.w2, 16 * kb), A.h1[kb], acc);
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] += L.b2[r];
        A.y0 = acc[0];                                   // row 0 on the g = 0 lanes
        if (L.g == 0) acc[0] = 0.0f;                     // slot of y0 in the colour input (its weight column is zero)
        A.x2[0] = pack8<0, false>(acc);
    }
    if (!CODED) encode_dir(d, L.g, A.x2[1], A.x2[2]);
    // L3: h2 = relu(W3 x2 + b3)   (b3 rides on the ones slot)
    if (PIN) asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        floatx16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) acc = mma32(lds_a(L.w3, t * 32 * LD3 + 16 * kb), A.x2[kb], acc);
        A.h2[2 * t] = pack8<0, true>(acc);
        A.h2[2 * t + 1] = pack8<8, true>(acc);
    }
    // L4: h3 = relu(W4 h2 + b4)
    if (PIN) asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        floatx16 acc = BIAS_LDS ? *reinterpret_cast<const floatx16*>(L.bias_lds + 64 + t * 32) : L.b4[t];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc = mma32(lds_a(L.w4, t * 32 * LD4 + 16 * kb), A.h2[kb], acc);
        A.h3[2 * t] = pack8<0, true>(acc);
        A.h3[2 * t + 1] = pack8<8, true>(acc);
    }
    // L5: rgb = sigmoid(W5 h3 + b5)  (rows 0..2 = registers 0..2 of the g = 0 lanes)
    if (PIN) asm volatile("" ::: "memory");
    {
        floatx16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc = mma32(lds_a(L.w5, 16 * kb), A.h3[kb], acc);
#pragma unroll
        for (int c = 0; c < 3; ++c) A.sg[c] = __builtin_amdgcn_rcpf(1.0f + __expf(-(acc[c] + L.b5[c])));
    }
}

template <typename TIO, bool NARROW>
DEV void fetch_inputs(const TIO* __restrict__ feats, const float* __restrict__ dirs, int64_t s, bool live, int g, int in_dim,
                      bf16x8 x0[2], float d[3]) {
    if (NARROW) {
        x0[0] = load_feats8_narrow<TIO>(feats + s * in_dim, 8 * g, in_dim, live);
        x0[1] = load_feats8_narrow<TIO>(feats + s * in_dim, 16 + 8 * g, in_dim, live);
    } else {
        x0[0] = load_feats8<TIO>(feats + s * IN + 8 * g, live);
        x0[1] = load_feats8<TIO>(feats + s * IN + 16 + 8 * g, live);
    }
    d[0] = live ? dirs[s * 3] : 0.0f; d[1] = live ? dirs[s * 3 + 1] : 0.0f; d[2] = live ? dirs[s * 3 + 2] : 0.0f;
}

// Per-RAY view code.  The view direction of a sample is the direction of its ray, and its encoding (3 sincos + 9 angle
// doublings + 16 selects + 8 packs: ~80 of the ~250 vector instructions of a forward tile) does not depend on the sample:
// dir_code_kernel encodes every ray once ([ray][g][16] bf16 = the two K blocks of both lane halves, 64 B) and the
// decoder kernels gather 32 bytes per lane by ray index.  Same arithmetic, same values.
__global__ void __launch_bounds__(256)
dir_code_kernel(const float* __restrict__ dirs, int64_t num_rays, bf16x8* __restrict__ code) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= num_rays * 2) return;
    const int64_t r = e >> 1;
    const int g = (int)(e & 1);
    const float d[3] = {dirs[r * 3], dirs[r * 3 + 1], dirs[r * 3 + 2]};
    bf16x8 k1, k2;
    encode_dir(d, g, k1, k2);
    code[e * 2] = k1;
    code[e * 2 + 1] = k2;
}
template <typename TIO, bool NARROW>
DEV void fetch_inputs_coded(const TIO* __restrict__ feats, const bf16x8* __restrict__ code, const int64_t* __restrict__ ridx,
                            int64_t s, bool live, int g, int in_dim, bf16x8 x0[2], bf16x8 k[2]) {
    if (NARROW) {
        x0[0] = load_feats8_narrow<TIO>(feats + s * in_dim, 8 * g, in_dim, live);
        x0[1] = load_feats8_narrow<TIO>(feats + s * in_dim, 16 + 8 * g, in_dim, live);
    } else {
        x0[0] = load_feats8<TIO>(feats + s * IN + 8 * g, live);
        x0[1] = load_feats8<TIO>(feats + s * IN + 16 + 8 * g, live);
    }
    const int64_t r = live ? ridx[s] : 0;
    const bf16x8* cp = code + (r * 2 + g) * 2;
    k[0] = cp[0]; k[1] = cp[1];
}

// ---------------------------------------------------------------------------------------------- forward kernel
constexpr int FWD_WAVES = 8;

constexpr int FWD_OFF_BIASV = L_FWD_END * 2;
constexpr int FWD_OFF_STG = FWD_OFF_BIASV + BIASV_FLOATS * 4;
constexpr int FWD_LDS = FWD_OFF_STG + NPARAM_PAD * 4;
template <typename TIO, bool NARROW, bool PIN, bool CODED = false>
__global__ void __launch_bounds__(FWD_WAVES * 64)
mlp_fwd_kernel(const TIO* __restrict__ feats, const float* __restrict__ dirs, const int64_t* __restrict__ ridx,
               int64_t num_samples, int in_dim,
               const float* __restrict__ params, float* __restrict__ out_rgb, float* __restrict__ out_density) {
    const bf16x8* code = reinterpret_cast<const bf16x8*>(dirs);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* sw = reinterpret_cast<__bf16*>(smem);
    float* biasv = reinterpret_cast<float*>(smem + FWD_OFF_BIASV);
    float* stg = reinterpret_cast<float*>(smem + FWD_OFF_STG);
    stage_params<FWD_WAVES * 64>(stg, params, threadIdx.x, NARROW ? in_dim : IN);
    __syncthreads();
    stage_weights<false>(sw, stg, threadIdx.x, FWD_WAVES * 64);
    if (PIN) stage_bias_vectors(biasv, stg, threadIdx.x, FWD_WAVES * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LaneConst L;
    lane_const<PIN>(L, sw, stg, lane, biasv);
    __syncthreads();

    const int64_t ntiles = (num_samples + TS - 1) / TS;
    const int64_t stride = (int64_t)gridDim.x * FWD_WAVES;
    int64_t tile = (int64_t)blockIdx.x * FWD_WAVES + wave;
    Acts A;
    float d[3] = {0.f, 0.f, 0.f};
    if (tile < ntiles) {
        if constexpr (CODED) fetch_inputs_coded<TIO, NARROW>(feats, code, ridx, tile * TS + L.n, tile * TS + L.n < num_samples, L.g, in_dim, A.x0, &A.x2[1]);
        else fetch_inputs<TIO, NARROW>(feats, dirs, tile * TS + L.n, tile * TS + L.n < num_samples, L.g, in_dim, A.x0, d);
    }
    for (; tile < ntiles; tile += stride) {
        const int64_t s = tile * TS + L.n;
        const bool live = s < num_samples;
        // prefetch the next tile's inputs behind this tile's arithmetic
        bf16x8 nx0[2], nk[2];
        float nd[3];
        const int64_t ns = (tile + stride) * TS + L.n;
        const bool more = tile + stride < ntiles;
        if (more) {
            if constexpr (CODED) fetch_inputs_coded<TIO, NARROW>(feats, code, ridx, ns, ns < num_samples, L.g, in_dim, nx0, nk);
            else fetch_inputs<TIO, NARROW>(feats, dirs, ns, ns < num_samples, L.g, in_dim, nx0, nd);
        }
        forward_tile<PIN, PIN, CODED>(L, d, A);
        if (L.g == 0 && live) {
            out_density[s] = fmaxf(A.y0, 0.0f);
            out_rgb[s * 3] = A.sg[0]; out_rgb[s * 3 + 1] = A.sg[1]; out_rgb[s * 3 + 2] = A.sg[2];
        }
        if (more) {
            A.x0[0] = nx0[0]; A.x0[1] = nx0[1];
            if constexpr (CODED) { A.x2[1] = nk[0]; A.x2[2] = nk[1]; }
            else { d[0] = nd[0]; d[1] = nd[1]; d[2] = nd[2]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward kernel
// Eight waves per workgroup in two roles, one of each per SIMD:
//   * chain waves 0..3 own a tile each: forward recompute, back-propagation through the transposed weights, grad_feats;
//     they publish the (dY, X) transposition images of every layer in LDS;
//   * accumulator waves 4..7 (partner of chain wave w is wave w + 4) read the images back transposed and keep
//     dW += dY^T X in registers for the whole launch (180 VGPRs).
// Splitting the roles keeps every wave at <= 256 registers: two waves per SIMD (the chain wave's conversions overlap
// the partner's MFMAs) and all MFMAs in VGPR form, where one combined wave needed ~480 registers, ran alone on its SIMD
// and paid a v_accvgpr_read for every accumulator value it converted.
// Each pair hands images over through a two-slot ring in LDS guarded by two counters (ready / done); no workgroup
// barrier in the main loop.
constexpr int BWD_PAIRS = 4;
constexpr int BWD_THREADS = 2 * BWD_PAIRS * 64;
constexpr int BWD_PAIR_LDS = 4 * TILE_BYTES;        // two slots x (dY image, X image)
constexpr int BWD_OFF_BIASV = L_BWD_END * 2;
constexpr int BWD_OFF_FLAGS = BWD_OFF_BIASV + BIASV_FLOATS * 4;
constexpr int BWD_OFF_IMG = BWD_OFF_FLAGS + 64;
constexpr int BWD_LDS_MAIN = BWD_OFF_IMG + BWD_PAIRS * BWD_PAIR_LDS;
constexpr int BWD_LDS_REDUCE = BWD_OFF_IMG + 2 * NPARAM_PAD * 4;      // 4 rows x half the parameters      // epilogue scratch aliases the images
constexpr int BWD_LDS = BWD_LDS_MAIN > BWD_LDS_REDUCE ? BWD_LDS_MAIN : BWD_LDS_REDUCE;
static_assert(BWD_LDS <= 160 * 1024, "LDS budget");
static_assert(NPARAM_PAD * 4 <= BWD_PAIRS * BWD_PAIR_LDS, "parameter staging aliases the images");

// dX block: acc[k][n] = sum over NKB chained K blocks of  WT[k][.] dY[n][.]
template <int NKB> DEV floatx16 back_block(const __bf16* wt_lane_row, const bf16x8* dy) {
    floatx16 acc = zero16();
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) acc = mma32(lds_a(wt_lane_row, 16 * kb), dy[kb], acc);
    return acc;
}

DEV void wait_at_least(int* counter, int target) {
    // the spin is bounded (~tens of ms) so that a protocol error shows up as wrong numbers in the tests, not as a hung GPU
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target && ++spins < (1 << 20))
        __builtin_amdgcn_s_sleep(1);
}
DEV void publish(int* counter, int value) { __hip_atomic_store(counter, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// weight-gradient accumulators of one accumulator wave
struct GradAcc {
    floatx4 dW5[4], dW4[16], dW3[12], dW2[4], dW1[8];
    floatx4 db;      // all bias gradients share ONE 16x16 block: row c = "combo" c (a 16-neuron block of one layer), see DB_*
};
constexpr int DB_L4 = 0, DB_L1 = 4, DB_L2 = 8, DB_L5 = 9;      // first combo row of each layer (b3 comes with dW3)
DEV void grad_zero(GradAcc& G) {
#pragma unroll
    for (int t = 0; t < 4; ++t) { G.dW5[t] = zero4(); G.dW2[t] = zero4(); }
#pragma unroll
    for (int t = 0; t < 16; ++t) G.dW4[t] = zero4();
#pragma unroll
    for (int t = 0; t < 12; ++t) G.dW3[t] = zero4();
#pragma unroll
    for (int t = 0; t < 8; ++t) G.dW1[t] = zero4();
    G.db = zero4();
}
// visit every (packed parameter index, accumulator element) pair this lane owns: f(index, value)
template <typename F> DEV void grad_visit(const GradAcc& G, int lane, F f) {
    const int col = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = 4 * rg + rr;                     // row inside a 16-row block
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (row < 3) f(OW5 + row * H + 16 * kt + col, G.dW5[kt][rr]);
            f(OW2 + row * H + 16 * kt + col, G.dW2[kt][rr]);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int R = 16 * it + row;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) f(OW4 + R * H + 16 * kt + col, G.dW4[it * 4 + kt][rr]);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) f(OW1 + R * IN + 16 * kt + col, G.dW1[it * 2 + kt][rr]);
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const int u = 16 * kt + col;             // feature index inside the 48-wide colour input
                if (kt == 0) { if (u >= 1) f(OW3 + R * X2 + u - 1, G.dW3[it * 3 + kt][rr]); }
                else if (u < ONES_SLOT) f(OW3 + R * X2 + u - 1, G.dW3[it * 3 + kt][rr]);
                else if (u == ONES_SLOT) f(OB3 + R, G.dW3[it * 3 + kt][rr]);
            }
        }
        {                                                // shared bias block: row = combo, column = neuron inside its block
            const int c = row;
            if (c < DB_L1) f(OB4 + 16 * (c - DB_L4) + col, G.db[rr]);
            else if (c < DB_L2) f(OB1 + 16 * (c - DB_L1) + col, G.db[rr]);
            else if (c == DB_L2) f(OB2 + col, G.db[rr]);
            else if (c == DB_L5 && col < 3) f(OB5 + col, G.db[rr]);
        }
    }
}

// one stage of an accumulator wave: dW[it][kt] += dY_it^T X_kt for the published images.  DB >= 0: the bias gradient
// sum_n dY[n][.] of block `it` is accumulated into row DB + it of the shared block by an MFMA whose A operand is all ones in
// that row and zero elsewhere (the transposed dY image is the B operand there).
template <int NIT, int NKT, int DB>
DEV void accumulate_stage(const unsigned char* imgY, const unsigned char* imgX, floatx4 (&dW)[NIT * NKT], floatx4& db, int lane) {
    bf16x8 xb[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) xb[kt] = load_transposed(imgX, kt);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const bf16x8 a = load_transposed(imgY, it);
        if (DB >= 0) {
            const unsigned one2 = (lane & 15) == DB + it ? 0x3f803f80u : 0u;      // two bf16 1.0
            const u32x4 row_of_ones = {one2, one2, one2, one2};
            db = mma16(__builtin_bit_cast(bf16x8, row_of_ones), a, db);
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) dW[it * NKT + kt] = mma16(a, xb[kt], dW[it * NKT + kt]);
    }
}

template <typename TIO, bool NARROW, bool CODED = false>
__global__ void __launch_bounds__(BWD_THREADS)
mlp_bwd_kernel(const TIO* __restrict__ feats, const float* __restrict__ dirs, const int64_t* __restrict__ ridx,
               int64_t num_samples, int in_dim,
               const float* __restrict__ params, const float* __restrict__ grad_rgb, const float* __restrict__ grad_density,
               TIO* __restrict__ grad_feats, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* sw = reinterpret_cast<__bf16*>(smem);
    float* biasv = reinterpret_cast<float*>(smem + BWD_OFF_BIASV);
    int* flags = reinterpret_cast<int*>(smem + BWD_OFF_FLAGS);              // ready[4], done[4]
    float* stg = reinterpret_cast<float*>(smem + BWD_OFF_IMG);              // aliases the images
    stage_params<BWD_THREADS>(stg, params, threadIdx.x, NARROW ? in_dim : IN);
    if (threadIdx.x < 16) flags[threadIdx.x] = 0;
    __syncthreads();
    stage_weights<true>(sw, stg, threadIdx.x, BWD_THREADS);
    stage_bias_vectors(biasv, stg, threadIdx.x, BWD_THREADS);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = wave & (BWD_PAIRS - 1);
    const bool chain_role = wave < BWD_PAIRS;
    LaneConst L;
    lane_const<true>(L, sw, stg, lane, biasv);
    __syncthreads();

    unsigned char* ring = smem + BWD_OFF_IMG + (size_t)pair * BWD_PAIR_LDS;
    int* ready = flags + pair;
    int* done = flags + BWD_PAIRS + pair;
    const int64_t ntiles = (num_samples + TS - 1) / TS;
    const int64_t stride = (int64_t)gridDim.x * BWD_PAIRS;
    int64_t tile = (int64_t)blockIdx.x * BWD_PAIRS + pair;
    int seq = 0;                                                             // stages handed over so far
    GradAcc G;

    if (chain_role) {
        const int n = L.n, g = L.g;
        const __bf16* w5t = sw + L_W5T + n * LT5 + 8 * g;                   // this lane's rows of the transposed operands
        const __bf16* w4t = sw + L_W4T + n * LT4 + 8 * g;
        const __bf16* w3t = sw + L_W3T + (n & 15) * LT3 + 8 * g;
        const __bf16* w2t = sw + L_W2T + n * LT2 + 8 * g;
        const __bf16* w1t = sw + L_W1T + n * LT1 + 8 * g;
        const int wc_off = (2 * n + g) * 8, wn_off = g * TILE_REGION + n * 16;
        Acts A;
        float d[3] = {0.f, 0.f, 0.f};
        const bf16x8* code = reinterpret_cast<const bf16x8*>(dirs);
        if (tile < ntiles) {
            if constexpr (CODED) fetch_inputs_coded<TIO, NARROW>(feats, code, ridx, tile * TS + n, tile * TS + n < num_samples, g, in_dim, A.x0, &A.x2[1]);
            else fetch_inputs<TIO, NARROW>(feats, dirs, tile * TS + n, tile * TS + n < num_samples, g, in_dim, A.x0, d);
        }
        for (; tile < ntiles; tile += stride) {
            const int64_t s = tile * TS + n;
            const bool live = s < num_samples;
            float gr[3] = {0.f, 0.f, 0.f}, gd = 0.0f;
            if (live && g == 0) { gr[0] = grad_rgb[s * 3]; gr[1] = grad_rgb[s * 3 + 1]; gr[2] = grad_rgb[s * 3 + 2]; gd = grad_density[s]; }
            bf16x8 nx0[2], nk[2];
            float nd[3];
            const int64_t ns = (tile + stride) * TS + n;
            const bool more = tile + stride < ntiles;
            if (more) {
                if constexpr (CODED) fetch_inputs_coded<TIO, NARROW>(feats, code, ridx, ns, ns < num_samples, g, in_dim, nx0, nk);
                else fetch_inputs<TIO, NARROW>(feats, dirs, ns, ns < num_samples, g, in_dim, nx0, nd);
            }

            forward_tile<true, false, CODED>(L, d, A);

            // ---- stage 5: dY5 = g_rgb * s (1 - s) (3 channels, natural slots 0..2 of the g = 0 lanes)
            float g5[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // the g = 1 lanes hold other rows in sg: keep them 0
#pragma unroll
            for (int c = 0; c < 3; ++c) g5[c] = g == 0 ? gr[c] * A.sg[c] * (1.0f - A.sg[c]) : 0.0f;
            const bf16x8 dy5 = pack8f(g5);
            {
                unsigned char* slot = ring + (seq & 1) * 2 * TILE_BYTES;
                wait_at_least(done, seq - 1);
                store_natural(slot + wn_off, 0, dy5);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) store_chained(slot + TILE_BYTES + wc_off, kb, A.h3[kb]);
                publish(ready, ++seq);
            }
            bf16x8 dh3[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {                    // dH3 = (W5^T dY5) * (h3 > 0)
                const floatx16 acc = back_block<1>(w5t + t * 32 * LT5, &dy5);
                dh3[2 * t] = pack8_masked<0>(acc, A.h3[2 * t]);
                dh3[2 * t + 1] = pack8_masked<8>(acc, A.h3[2 * t + 1]);
            }
            // ---- stage 4
            {
                unsigned char* slot = ring + (seq & 1) * 2 * TILE_BYTES;
                wait_at_least(done, seq - 1);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) { store_chained(slot + wc_off, kb, dh3[kb]); store_chained(slot + TILE_BYTES + wc_off, kb, A.h2[kb]); }
                publish(ready, ++seq);
            }
            bf16x8 dh2[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {                    // dH2 = (W4^T dH3) * (h2 > 0)
                const floatx16 acc = back_block<4>(w4t + t * 32 * LT4, dh3);
                dh2[2 * t] = pack8_masked<0>(acc, A.h2[2 * t]);
                dh2[2 * t + 1] = pack8_masked<8>(acc, A.h2[2 * t + 1]);
            }
            // ---- stage 3
            {
                unsigned char* slot = ring + (seq & 1) * 2 * TILE_BYTES;
                wait_at_least(done, seq - 1);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) store_chained(slot + wc_off, kb, dh2[kb]);
                store_chained(slot + TILE_BYTES + wc_off, 0, A.x2[0]);
                store_natural(slot + TILE_BYTES + wn_off, 1, A.x2[1]);
                store_natural(slot + TILE_BYTES + wn_off, 2, A.x2[2]);
                publish(ready, ++seq);
            }
            bf16x8 dy2;
            {                                                // dY2[m] = W3^T dH2 (m = 1..15), dY2[0] = g_density * (y0 > 0)
                floatx16 acc = back_block<4>(w3t, dh2);
                if (g == 0) acc[0] = A.y0 > 0.0f ? gd : 0.0f;
                dy2 = pack8<0, false>(acc);
            }
            // ---- stage 2
            {
                unsigned char* slot = ring + (seq & 1) * 2 * TILE_BYTES;
                wait_at_least(done, seq - 1);
                store_chained(slot + wc_off, 0, dy2);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) store_chained(slot + TILE_BYTES + wc_off, kb, A.h1[kb]);
                publish(ready, ++seq);
            }
            bf16x8 dh1[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {                    // dH1 = (W2^T dY2) * (h1 > 0)
                const floatx16 acc = back_block<1>(w2t + t * 32 * LT2, &dy2);
                dh1[2 * t] = pack8_masked<0>(acc, A.h1[2 * t]);
                dh1[2 * t + 1] = pack8_masked<8>(acc, A.h1[2 * t + 1]);
            }
            // ---- stage 1
            {
                unsigned char* slot = ring + (seq & 1) * 2 * TILE_BYTES;
                wait_at_least(done, seq - 1);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) store_chained(slot + wc_off, kb, dh1[kb]);
                store_natural(slot + TILE_BYTES + wn_off, 0, A.x0[0]);
                store_natural(slot + TILE_BYTES + wn_off, 1, A.x0[1]);
                publish(ready, ++seq);
            }
            {                                                // dX0 = W1^T dH1 -> grad_feats
                const floatx16 acc = back_block<4>(w1t, dh1);
                if (live) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (NARROW)
                            store_grad4_narrow<TIO>(grad_feats + s * in_dim, 8 * q + 4 * g, in_dim, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                        else
                            store_grad4<TIO>(grad_feats + s * IN + 8 * q + 4 * g, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                    }
                }
            }
            if (more) {
                A.x0[0] = nx0[0]; A.x0[1] = nx0[1];
                if constexpr (CODED) { A.x2[1] = nk[0]; A.x2[2] = nk[1]; }
                else { d[0] = nd[0]; d[1] = nd[1]; d[2] = nd[2]; }
            }
        }
    } else {
        grad_zero(G);
        const int tr_off = ((lane >> 1) & 1) * TILE_REGION + (8 * (lane >> 4) + 2 * ((lane >> 2) & 3) + (lane & 1)) * 8;
        for (; tile < ntiles; tile += stride) {
            const unsigned char* slot;
#define WISP_NEXT_STAGE()  slot = ring + (seq & 1) * 2 * TILE_BYTES + tr_off; wait_at_least(ready, seq + 1)
            WISP_NEXT_STAGE(); accumulate_stage<1, 4, DB_L5>(slot, slot + TILE_BYTES, G.dW5, G.db, lane); publish(done, ++seq);
            WISP_NEXT_STAGE(); accumulate_stage<4, 4, DB_L4>(slot, slot + TILE_BYTES, G.dW4, G.db, lane); publish(done, ++seq);
            WISP_NEXT_STAGE(); accumulate_stage<4, 3, -1>(slot, slot + TILE_BYTES, G.dW3, G.db, lane); publish(done, ++seq);
            WISP_NEXT_STAGE(); accumulate_stage<1, 4, DB_L2>(slot, slot + TILE_BYTES, G.dW2, G.db, lane); publish(done, ++seq);
            WISP_NEXT_STAGE(); accumulate_stage<4, 2, DB_L1>(slot, slot + TILE_BYTES, G.dW1, G.db, lane); publish(done, ++seq);
#undef WISP_NEXT_STAGE
        }
    }

    // ---- epilogue: one partial row per workgroup.  In two halves of the parameter range (four full rows do not fit in
    // LDS next to the weights) the accumulator waves park their registers in LDS, then all threads sum the four rows
    // and write the result with coalesced stores.
    constexpr int HALF = NPARAM_PAD / 2;
    float* red = reinterpret_cast<float*>(smem + BWD_OFF_IMG);              // [4][HALF], aliases the images
    float* out = partials + (int64_t)blockIdx.x * NPARAM_PAD;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (!chain_role) {
            float* r = red + pair * HALF - h * HALF;
            grad_visit(G, lane, [&](int idx, float v) { if ((idx >= HALF) == (h == 1)) r[idx] = v; });
        }
        __syncthreads();
        for (int j = threadIdx.x; j < HALF; j += BWD_THREADS)
            out[h * HALF + j] = (red[j] + red[HALF + j]) + (red[2 * HALF + j] + red[3 * HALF + j]);
    }
}

int cu_count() {
    static int n = [] { int d = 0, c = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d); return c > 0 ? c : 256; }();
    return n;
}

template <typename TIO, bool NARROW, bool CODED>
int launch_fwd(const void* feats, const float* dirs, const int64_t* ridx, int64_t S, int in_dim, const float* params, float* rgb,
               float* density, hipStream_t st) {
    const size_t lds = FWD_LDS;
    const int64_t ntiles = (S + TS - 1) / TS;
    static const int pin = [] { const char* v = getenv("WISP_MLP_FWD_PIN"); return v && v[0] ? atoi(v) : 2; }();   // workgroups per CU; 0 = register-resident weights
    if (pin > 0 || CODED) {
        auto kern = mlp_fwd_kernel<TIO, NARROW, true, CODED>;
        const hipError_t e = WISP_ALLOW_LDS(kern, lds);
        if (e != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, "nerf_mlp_bf16", hipGetErrorString(e));
        const int grid = (int)min64(ceil_div64(ntiles, FWD_WAVES), (int64_t)(pin > 0 ? pin : 2) * cu_count());
        hipLaunchKernelGGL(kern, dim3(grid), dim3(FWD_WAVES * 64), lds, st, (const TIO*)feats, dirs, ridx, S, in_dim, params, rgb, density);
    } else if constexpr (!CODED) {
        auto kern = mlp_fwd_kernel<TIO, NARROW, false, false>;
        const hipError_t e = WISP_ALLOW_LDS(kern, lds);
        if (e != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, "nerf_mlp_bf16", hipGetErrorString(e));
        const int grid = (int)min64(ceil_div64(ntiles, FWD_WAVES), cu_count());
        hipLaunchKernelGGL(kern, dim3(grid), dim3(FWD_WAVES * 64), lds, st, (const TIO*)feats, dirs, ridx, S, in_dim, params, rgb, density);
    }
    return 0;
}

template <typename TIO, bool NARROW, bool CODED>
int launch_bwd(const void* feats, const float* dirs, const int64_t* ridx, int64_t S, int in_dim, const float* params,
               const float* grad_rgb, const float* grad_density, void* grad_feats, float* partials, int* partial_rows, hipStream_t st) {
    const size_t lds = BWD_LDS;
    auto kern = mlp_bwd_kernel<TIO, NARROW, CODED>;
    const hipError_t e = WISP_ALLOW_LDS(kern, lds);
    if (e != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, "nerf_mlp_bf16", hipGetErrorString(e));
    const int64_t ntiles = (S + TS - 1) / TS;
    const int grid = (int)min64(ceil_div64(ntiles, BWD_PAIRS), cu_count());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BWD_THREADS), lds, st, (const TIO*)feats, dirs, ridx, S, in_dim, params, grad_rgb,
                       grad_density, (TIO*)grad_feats, partials);
    *partial_rows = grid;
    return 0;
}

}  // namespace

namespace wisp_mlp {

#define WISP_MLP_IO(FN, ...)                                                                        \
    if (in_dim == IN) switch (dtype_io) {                                                           \
        case WISP_F32: return FN<float, false, false>(__VA_ARGS__);                                 \
        case WISP_F16: return FN<__half, false, false>(__VA_ARGS__);                                \
        default: return FN<__hip_bfloat16, false, false>(__VA_ARGS__);                              \
    }                                                                                               \
    switch (dtype_io) {                                                                             \
        case WISP_F32: return FN<float, true, false>(__VA_ARGS__);                                  \
        case WISP_F16: return FN<__half, true, false>(__VA_ARGS__);                                 \
        default: return FN<__hip_bfloat16, true, false>(__VA_ARGS__);                               \
    }

int bf16_forward(const void* feats, int dtype_io, const float* dirs, int64_t S, int in_dim, const float* params, float* rgb,
                 float* density, hipStream_t st) {
    WISP_MLP_IO(launch_fwd, feats, dirs, nullptr, S, in_dim, params, rgb, density, st)
}

int bf16_backward(const void* feats, int dtype_io, const float* dirs, int64_t S, int in_dim, const float* params,
                  const float* grad_rgb, const float* grad_density, void* grad_feats, float* partials, int* partial_rows,
                  hipStream_t st) {
    WISP_MLP_IO(launch_bwd, feats, dirs, nullptr, S, in_dim, params, grad_rgb, grad_density, grad_feats, partials, partial_rows, st)
}
#undef WISP_MLP_IO

// per-ray view code variants: every feature width and I/O type the per-sample kernels take (round 4: the narrow fp32 rows of the
// octree / codebook fields too - their decoder launches paid the in-kernel sin / cos encoding of 2 M directions twice per step)
bool bf16_rays_supported(int dtype_io, int in_dim) {
    return in_dim >= 1 && in_dim <= IN && (dtype_io == WISP_F32 || dtype_io == WISP_F16 || dtype_io == WISP_BF16);
}

void bf16_dir_code(const float* dirs, int64_t num_rays, void* code, hipStream_t st) {
    if (num_rays <= 0) return;
    hipLaunchKernelGGL(dir_code_kernel, dim3((unsigned)ceil_div64(num_rays * 2, 256)), dim3(256), 0, st, dirs, num_rays,
                       reinterpret_cast<bf16x8*>(code));
}

#define WISP_MLP_IO_CODED(FN, ...)                                                                  \
    if (in_dim == IN) switch (dtype_io) {                                                           \
        case WISP_F32: return FN<float, false, true>(__VA_ARGS__);                                  \
        case WISP_F16: return FN<__half, false, true>(__VA_ARGS__);                                 \
        default: return FN<__hip_bfloat16, false, true>(__VA_ARGS__);                               \
    }                                                                                               \
    switch (dtype_io) {                                                                             \
        case WISP_F32: return FN<float, true, true>(__VA_ARGS__);                                   \
        case WISP_F16: return FN<__half, true, true>(__VA_ARGS__);                                  \
        default: return FN<__hip_bfloat16, true, true>(__VA_ARGS__);                                \
    }

int bf16_forward_rays(const void* feats, int dtype_io, const void* code, const int64_t* ridx, int64_t S, int in_dim,
                      const float* params, float* rgb, float* density, hipStream_t st) {
    const float* c = reinterpret_cast<const float*>(code);
    WISP_MLP_IO_CODED(launch_fwd, feats, c, ridx, S, in_dim, params, rgb, density, st)
}

int bf16_backward_rays(const void* feats, int dtype_io, const void* code, const int64_t* ridx, int64_t S, int in_dim,
                       const float* params, const float* grad_rgb, const float* grad_density, void* grad_feats, float* partials,
                       int* partial_rows, hipStream_t st) {
    const float* c = reinterpret_cast<const float*>(code);
    WISP_MLP_IO_CODED(launch_bwd, feats, c, ridx, S, in_dim, params, grad_rgb, grad_density, grad_feats, partials, partial_rows, st)
}
#undef WISP_MLP_IO_CODED

}  // namespace wisp_mlp
