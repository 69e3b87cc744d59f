// Fused radiance-field decoder for WIDE hidden layers (hidden = 128; written for any multiple of 64 whose weight images
// fit LDS) on the matrix cores.
//
// The reference's best configuration, the documented VQAD command line and NeuralSDF all use hidden_dim = 128
// (docs/pages/app_nerf.md:185-192, wisp/models/nefs/nerf.py:151-173); nerf_mlp_bf16.hip is hand-scheduled for 64.  This
// file keeps that kernel's central idea - activations stay in registers between layers: sample n = lane & 31 sits on the
// N side of v_mfma_f32_32x32x16_bf16, the accumulator registers of a 32-row block ARE two K blocks of the next layer after
// one v_cvt_pk_bf16_f32 per pair, the weight columns in LDS are stored pre-permuted to match - written as loops over
// NB = hidden / 32 row blocks and NK = hidden / 16 K blocks.  What does NOT carry over is the weight-gradient scheme: at
// hidden 128 the dW blocks (120 of 16x16) no longer fit the registers of one accumulator wave next to 125 KB of weight
// images, so the backward is split in two kernels:
//   * chain kernel: forward recompute, back-propagation through the transposed (also permuted) weights, grad_feats; every
//     (dY, X) operand pair of the five weight gradients is dumped to a scratch buffer exactly as it sits in the registers
//     (one coalesced 1 KB store per K block and wave);
//   * dW kernel: four waves load four tiles' dY operands back, recompute the activations, write both into LDS transposition images
//     (ds_read_b64_tr_b16 turns "lane = sample" into "K = sample"), and accumulate dW += dY^T X with
//     v_mfma_f32_16x16x32_bf16 - each wave owns a quarter of the 16x16 blocks for the whole launch (120 VGPRs) and adds
//     them to the workgroup's partial row at the end; a small kernel sums the rows.
// Only the dY operands go through the scratch (26 KB per 32-sample tile = 0.83 KB per sample, written once and read once);
// the X operands - the activations - are re-derived in the dW kernel by running the tile through the forward pass again
// (68 MFMAs), which is cheaper than another 0.9 KB per sample of traffic.
#include "nerf_mlp_bf16_dev.h"
#include <cstdlib>

namespace {
using namespace wisp_mlp_dev;

template <int HH> struct Wide {
    static_assert(HH % 64 == 0 && HH >= 64, "hidden width: a multiple of 64");
    static constexpr int NB = HH / 32, NK = HH / 16;
    // canonical parameter order  W1[H,IN] b1[H] W2[16,H] b2[16] W3[H,X2] b3[H] W4[H,H] b4[H] W5[3,H] b5[3]
    static constexpr int OW1 = 0, OB1 = OW1 + HH * IN, OW2 = OB1 + HH, OB2 = OW2 + 16 * HH, OW3 = OB2 + 16, OB3 = OW3 + HH * X2,
                         OW4 = OB3 + HH, OB4 = OW4 + HH * HH, OW5 = OB4 + HH, OB5 = OW5 + 3 * HH, NPARAM = OB5 + 3;
    static constexpr int NPARAM_PAD = (NPARAM + 63) / 64 * 64;
    // forward operand images [out row][K slots], row stride = K + 8 elements (conflict-free ds_read_b128)
    static constexpr int LD1 = IN + 8, LD2 = HH + 8, LD3 = 48 + 8, LD4 = HH + 8, LD5 = HH + 8;
    static constexpr int L_W1 = 0, L_W2 = L_W1 + HH * LD1, L_W3 = L_W2 + 16 * LD2, L_W4 = L_W3 + HH * LD3, L_W5 = L_W4 + HH * LD4,
                         L_FWD_END = L_W5 + 4 * LD5;
    // backward operand images [in row][out-neuron slots]
    static constexpr int LT5 = 24, LT4 = HH + 8, LT3 = HH + 8, LT2 = 24, LT1 = HH + 8;
    static constexpr int L_W5T = L_FWD_END, L_W4T = L_W5T + HH * LT5, L_W3T = L_W4T + HH * LT4, L_W2T = L_W3T + 16 * LT3,
                         L_W1T = L_W2T + HH * LT2, L_BWD_END = L_W1T + IN * LT1;
    static constexpr int BIASV_FLOATS = 2 * NB * 2 * 16;          // [layer 1 | layer 4][t][g][16] in accumulator layout
    // scratch slots of one tile (1 KB each: 64 lanes x 16 B): the dY operand of every weight gradient, in the order the dW
    // kernel consumes them (the X operands are re-derived there by a forward pass: cheaper than 0.9 KB / sample of traffic)
    static constexpr int S5Y = 0, S4Y = S5Y + 1, S3Y = S4Y + NK, S2Y = S3Y + NK, S1Y = S2Y + 1, NSLOT = S1Y + NK;
    static constexpr int IMG_BYTES = (HH / 8) * TILE_REGION;      // transposition image of HH features x 32 samples
    // forward + transposed weight images must fit the 160 KB of LDS next to the bias vectors: true for 64 and 128
    static_assert((L_BWD_END * 2 + BIASV_FLOATS * 4) <= 160 * 1024, "weight images exceed the LDS of one CU");
};

__host__ __device__ inline int wide_packed_index(int canonical, int in_dim, int hh) {
    if (canonical < hh * IN) {
        const int r = canonical / IN, c = canonical % IN;
        return c < in_dim ? r * in_dim + c : -1;
    }
    return canonical - hh * (IN - in_dim);
}
DEV float wide_param(const float* __restrict__ P, int canonical, int in_dim, int hh) {
    const int i = wide_packed_index(canonical, in_dim, hh);
    return i < 0 ? 0.0f : P[i];
}

// Weights from the caller's packed buffer (W1 rows in_dim wide) into the permuted bf16 operand images, once per workgroup.
// The parameter vector is read in ITS order - consecutive threads, consecutive floats, several loads in flight - and every
// value is scattered to its slot(s): phi16 is an involution, so "slot of column c" is phi(c) just like "column of slot s".
// (Walking the images and gathering the parameters instead cost a scattered global load plus two integer divisions per
// element: ~0.1 ms per workgroup and launch, more than the forward pass itself.)
template <int HH, bool BWD>
DEV void stage_weights_wide(__bf16* sw, float* biasv, const float* __restrict__ P, int in_dim, int tid, int nthreads) {
    typedef Wide<HH> W;
    {   // zero everything first: padding columns (in_dim < 32, the y0 slot, slots 44-47, row 3 of W5, unused W3T / W5T slots)
        uint4* z = reinterpret_cast<uint4*>(sw);
        const int n16 = ((BWD ? W::L_BWD_END : W::L_FWD_END) * 2) / 16;
        for (int e = tid; e < n16; e += nthreads) z[e] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    // accumulator-layout position of neuron h inside a bias vector: block t = h / 32, row rho = h % 32 = acc_row(r, g)
    auto bias_pos = [](int h) { const int t = h >> 5, rho = h & 31; return ((rho & 3) + 4 * (rho >> 3)) + 16 * ((rho >> 2) & 1) + 32 * t; };
    const float* p = P;
#pragma unroll 4
    for (int e = tid; e < HH * in_dim; e += nthreads) {                        // W1 [HH, in_dim]
        const int r = e / in_dim, c = e - r * in_dim;
        const __bf16 v = (__bf16)p[e];
        sw[W::L_W1 + r * W::LD1 + c] = v;
        if (BWD) sw[W::L_W1T + c * W::LT1 + phi(r)] = v;
    }
    p += HH * in_dim;
    for (int e = tid; e < HH; e += nthreads) biasv[bias_pos(e)] = p[e];        // b1
    p += HH;
#pragma unroll 4
    for (int e = tid; e < 16 * HH; e += nthreads) {                            // W2 [16, HH]
        const int r = e / HH, c = e - r * HH;
        const __bf16 v = (__bf16)p[e];
        sw[W::L_W2 + r * W::LD2 + phi(c)] = v;
        if (BWD) sw[W::L_W2T + c * W::LT2 + phi16(r)] = v;
    }
    p += 16 * HH + 16;                                                         // (b2 is read by lane_const_wide)
#pragma unroll 4
    for (int e = tid; e < HH * X2; e += nthreads) {                            // W3 [HH, 42]: column c <-> colour input u = c + 1
        const int r = e / X2, c = e - r * X2, u = c + 1;
        const __bf16 v = (__bf16)p[e];
        sw[W::L_W3 + r * W::LD3 + (u < 16 ? phi16(u) : u)] = v;
        if (BWD && u < 16) sw[W::L_W3T + u * W::LT3 + phi(r)] = v;
    }
    p += HH * X2;
    for (int e = tid; e < HH; e += nthreads) sw[W::L_W3 + e * W::LD3 + ONES_SLOT] = (__bf16)p[e];     // b3 rides on the ones slot
    p += HH;
#pragma unroll 4
    for (int e = tid; e < HH * HH; e += nthreads) {                            // W4 [HH, HH]
        const int r = e / HH, c = e - r * HH;
        const __bf16 v = (__bf16)p[e];
        sw[W::L_W4 + r * W::LD4 + phi(c)] = v;
        if (BWD) sw[W::L_W4T + c * W::LT4 + phi(r)] = v;
    }
    p += HH * HH;
    for (int e = tid; e < HH; e += nthreads) biasv[32 * W::NB + bias_pos(e)] = p[e];      // b4
    p += HH;
    for (int e = tid; e < 3 * HH; e += nthreads) {                             // W5 [3, HH]
        const int r = e / HH, c = e - r * HH;
        const __bf16 v = (__bf16)p[e];
        sw[W::L_W5 + r * W::LD5 + phi(c)] = v;
        if (BWD) sw[W::L_W5T + c * W::LT5 + r] = v;
    }
}

// The weight images never change after staging, so the optimiser would hoist every ds_read of an A operand out of the tile
// loop and try to keep 125 KB of weights in registers (measured: 496-512 VGPRs and spills).  A compiler-level memory
// barrier per layer keeps the reads where they are used.
DEV void keep_lds_reads_here() { asm volatile("" ::: "memory"); }

template <int HH> struct ActsW {
    bf16x8 x0[2], h1[Wide<HH>::NK], x2[3], h2[Wide<HH>::NK], h3[Wide<HH>::NK];
    float y0, sg[3];
};

template <int HH> struct LaneW {
    int n, g;
    const __bf16 *w1, *w2, *w3, *w4, *w5;
    const float* bias_lds;           // + 16 g
    float b2[8], b5[3];
};

template <int HH>
DEV void lane_const_wide(LaneW<HH>& L, const __bf16* sw, const float* biasv, const float* __restrict__ P, int in_dim, int lane) {
    typedef Wide<HH> W;
    L.n = lane & 31; L.g = lane >> 5;
    L.w1 = sw + W::L_W1 + L.n * W::LD1 + 8 * L.g;
    L.w2 = sw + W::L_W2 + (L.n & 15) * W::LD2 + 8 * L.g;
    L.w3 = sw + W::L_W3 + L.n * W::LD3 + 8 * L.g;
    L.w4 = sw + W::L_W4 + L.n * W::LD4 + 8 * L.g;
    L.w5 = sw + W::L_W5 + (L.n < 3 ? L.n : 3) * W::LD5 + 8 * L.g;
    L.bias_lds = biasv + 16 * L.g;
    const float* pb2 = P + HH * in_dim + HH + 16 * HH;                        // packed offsets (W1 rows in_dim wide)
    const float* pb5 = pb2 + 16 + HH * X2 + HH + HH * HH + HH + 3 * HH;
#pragma unroll
    for (int r = 0; r < 8; ++r) L.b2[r] = pb2[acc_row(r, L.g)];
#pragma unroll
    for (int c = 0; c < 3; ++c) L.b5[c] = pb5[c];
}

template <int HH>
DEV void forward_tile_wide(const LaneW<HH>& L, const float d[3], ActsW<HH>& A) {
    typedef Wide<HH> W;
    // L1: h1 = relu(W1 x0 + b1)
    keep_lds_reads_here();
#pragma unroll
    for (int t = 0; t < W::NB; ++t) {
        floatx16 acc = *reinterpret_cast<const floatx16*>(L.bias_lds + t * 32);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc = mma32(lds_a(L.w1, t * 32 * W::LD1 + 16 * kb), A.x0[kb], acc);
        A.h1[2 * t] = pack8<0, true>(acc);
        A.h1[2 * t + 1] = pack8<8, true>(acc);
    }
    // L2: y = W2 h1 + b2 (16 rows = accumulator registers 0..7); density = relu(y0)
    keep_lds_reads_here();
    {
        floatx16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < W::NK; ++kb) acc = mma32(lds_a(L.w2, 16 * kb), A.h1[kb], acc);
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] += L.b2[r];
        A.y0 = acc[0];
        if (L.g == 0) acc[0] = 0.0f;
        A.x2[0] = pack8<0, false>(acc);
    }
    encode_dir(d, L.g, A.x2[1], A.x2[2]);
    // L3: h2 = relu(W3 x2 + b3)   (b3 rides on the ones slot)
    keep_lds_reads_here();
#pragma unroll
    for (int t = 0; t < W::NB; ++t) {
        floatx16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) acc = mma32(lds_a(L.w3, t * 32 * W::LD3 + 16 * kb), A.x2[kb], acc);
        A.h2[2 * t] = pack8<0, true>(acc);
        A.h2[2 * t + 1] = pack8<8, true>(acc);
    }
    // L4: h3 = relu(W4 h2 + b4)
#pragma unroll
    for (int t = 0; t < W::NB; ++t) {
        keep_lds_reads_here();
        floatx16 acc = *reinterpret_cast<const floatx16*>(L.bias_lds + 32 * W::NB + t * 32);
#pragma unroll
        for (int kb = 0; kb < W::NK; ++kb) acc = mma32(lds_a(L.w4, t * 32 * W::LD4 + 16 * kb), A.h2[kb], acc);
        A.h3[2 * t] = pack8<0, true>(acc);
        A.h3[2 * t + 1] = pack8<8, true>(acc);
    }
    keep_lds_reads_here();
    // L5: rgb = sigmoid(W5 h3 + b5)
    {
        floatx16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < W::NK; ++kb) acc = mma32(lds_a(L.w5, 16 * kb), A.h3[kb], acc);
#pragma unroll
        for (int c = 0; c < 3; ++c) A.sg[c] = __builtin_amdgcn_rcpf(1.0f + __expf(-(acc[c] + L.b5[c])));
    }
}

template <typename TIO>
DEV void fetch_inputs_wide(const TIO* __restrict__ feats, const float* __restrict__ dirs, int64_t s, bool live, int g, int in_dim,
                           bf16x8 x0[2], float d[3]) {
    if (in_dim != IN) {
        x0[0] = load_feats8_narrow<TIO>(feats + s * in_dim, 8 * g, in_dim, live);
        x0[1] = load_feats8_narrow<TIO>(feats + s * in_dim, 16 + 8 * g, in_dim, live);
    } else {
        x0[0] = load_feats8<TIO>(feats + s * IN + 8 * g, live);
        x0[1] = load_feats8<TIO>(feats + s * IN + 16 + 8 * g, live);
    }
    d[0] = live ? dirs[s * 3] : 0.0f; d[1] = live ? dirs[s * 3 + 1] : 0.0f; d[2] = live ? dirs[s * 3 + 2] : 0.0f;
}

// ---------------------------------------------------------------------------------------------- forward kernel
constexpr int WF_WAVES = 8;

template <int HH, typename TIO>
__global__ void __launch_bounds__(WF_WAVES * 64)
wide_fwd_kernel(const TIO* __restrict__ feats, const float* __restrict__ dirs, int64_t num_samples, int in_dim,
                const float* __restrict__ params, float* __restrict__ out_rgb, float* __restrict__ out_density) {
    typedef Wide<HH> W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* sw = reinterpret_cast<__bf16*>(smem);
    float* biasv = reinterpret_cast<float*>(smem + (size_t)W::L_FWD_END * 2);
    stage_weights_wide<HH, false>(sw, biasv, params, in_dim, threadIdx.x, WF_WAVES * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LaneW<HH> L;
    lane_const_wide<HH>(L, sw, biasv, params, in_dim, lane);
    __syncthreads();
    const int64_t ntiles = (num_samples + TS - 1) / TS;
    const int64_t stride = (int64_t)gridDim.x * WF_WAVES;
    ActsW<HH> A;
    float d[3];
    for (int64_t tile = (int64_t)blockIdx.x * WF_WAVES + wave; tile < ntiles; tile += stride) {
        const int64_t s = tile * TS + L.n;
        const bool live = s < num_samples;
        fetch_inputs_wide<TIO>(feats, dirs, s, live, L.g, in_dim, A.x0, d);
        forward_tile_wide<HH>(L, d, A);
        if (L.g == 0 && live) {
            out_density[s] = fmaxf(A.y0, 0.0f);
            out_rgb[s * 3] = A.sg[0]; out_rgb[s * 3 + 1] = A.sg[1]; out_rgb[s * 3 + 2] = A.sg[2];
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward: chain kernel
constexpr int WC_WAVES = 8;

template <int NKB> DEV floatx16 back_block_w(const __bf16* wt_lane_row, const bf16x8* dy) {
    keep_lds_reads_here();
    floatx16 acc = zero16();
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) acc = mma32(lds_a(wt_lane_row, 16 * kb), dy[kb], acc);
    return acc;
}

template <int HH, typename TIO>
__global__ void __launch_bounds__(WC_WAVES * 64)
wide_chain_kernel(const TIO* __restrict__ feats, const float* __restrict__ dirs, int64_t num_samples, int in_dim,
                  const float* __restrict__ params, const float* __restrict__ grad_rgb, const float* __restrict__ grad_density,
                  TIO* __restrict__ grad_feats, bf16x8* __restrict__ scratch) {
    typedef Wide<HH> W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* sw = reinterpret_cast<__bf16*>(smem);
    float* biasv = reinterpret_cast<float*>(smem + (size_t)W::L_BWD_END * 2);
    stage_weights_wide<HH, true>(sw, biasv, params, in_dim, threadIdx.x, WC_WAVES * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LaneW<HH> L;
    lane_const_wide<HH>(L, sw, biasv, params, in_dim, lane);
    __syncthreads();
    const int n = L.n, g = L.g;
    const __bf16* w5t = sw + W::L_W5T + n * W::LT5 + 8 * g;
    const __bf16* w4t = sw + W::L_W4T + n * W::LT4 + 8 * g;
    const __bf16* w3t = sw + W::L_W3T + (n & 15) * W::LT3 + 8 * g;
    const __bf16* w2t = sw + W::L_W2T + n * W::LT2 + 8 * g;
    const __bf16* w1t = sw + W::L_W1T + n * W::LT1 + 8 * g;
    const int64_t ntiles = (num_samples + TS - 1) / TS;
    const int64_t stride = (int64_t)gridDim.x * WC_WAVES;
    ActsW<HH> A;
    float d[3];
    for (int64_t tile = (int64_t)blockIdx.x * WC_WAVES + wave; tile < ntiles; tile += stride) {
        const int64_t s = tile * TS + n;
        const bool live = s < num_samples;
        float gr[3] = {0.f, 0.f, 0.f}, gd = 0.0f;
        if (live && g == 0) { gr[0] = grad_rgb[s * 3]; gr[1] = grad_rgb[s * 3 + 1]; gr[2] = grad_rgb[s * 3 + 2]; gd = grad_density[s]; }
        fetch_inputs_wide<TIO>(feats, dirs, s, live, g, in_dim, A.x0, d);
        forward_tile_wide<HH>(L, d, A);
        bf16x8* out = scratch + tile * (int64_t)(W::NSLOT * 64) + lane;        // slot k of this tile: out[k * 64]
        // ---- stage 5: dY5 = g_rgb * s (1 - s)  (natural slots 0..2 of the g = 0 lanes) ; X = h3
        float g5[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) g5[c] = g == 0 ? gr[c] * A.sg[c] * (1.0f - A.sg[c]) : 0.0f;
        const bf16x8 dy5 = pack8f(g5);
        out[W::S5Y * 64] = dy5;
        bf16x8 dh3[W::NK];
#pragma unroll
        for (int t = 0; t < W::NB; ++t) {                       // dH3 = (W5^T dY5) * (h3 > 0)
            const floatx16 acc = back_block_w<1>(w5t + t * 32 * W::LT5, &dy5);
            dh3[2 * t] = pack8_masked<0>(acc, A.h3[2 * t]);
            dh3[2 * t + 1] = pack8_masked<8>(acc, A.h3[2 * t + 1]);
        }
        // ---- stage 4: dY = dH3, X = h2
#pragma unroll
        for (int kb = 0; kb < W::NK; ++kb) out[(W::S4Y + kb) * 64] = dh3[kb];
        bf16x8 dh2[W::NK];
#pragma unroll
        for (int t = 0; t < W::NB; ++t) {                       // dH2 = (W4^T dH3) * (h2 > 0)
            const floatx16 acc = back_block_w<W::NK>(w4t + t * 32 * W::LT4, dh3);
            dh2[2 * t] = pack8_masked<0>(acc, A.h2[2 * t]);
            dh2[2 * t + 1] = pack8_masked<8>(acc, A.h2[2 * t + 1]);
        }
        // ---- stage 3: dY = dH2, X = x2 (chained density-feature block + two natural view-encoding blocks)
#pragma unroll
        for (int kb = 0; kb < W::NK; ++kb) out[(W::S3Y + kb) * 64] = dh2[kb];
        bf16x8 dy2;
        {                                                       // dY2[m] = W3^T dH2 (m = 1..15), dY2[0] = g_density * (y0 > 0)
            floatx16 acc = back_block_w<W::NK>(w3t, dh2);
            if (g == 0) acc[0] = A.y0 > 0.0f ? gd : 0.0f;
            dy2 = pack8<0, false>(acc);
        }
        // ---- stage 2: dY = dY2, X = h1
        out[W::S2Y * 64] = dy2;
        bf16x8 dh1[W::NK];
#pragma unroll
        for (int t = 0; t < W::NB; ++t) {                       // dH1 = (W2^T dY2) * (h1 > 0)
            const floatx16 acc = back_block_w<1>(w2t + t * 32 * W::LT2, &dy2);
            dh1[2 * t] = pack8_masked<0>(acc, A.h1[2 * t]);
            dh1[2 * t + 1] = pack8_masked<8>(acc, A.h1[2 * t + 1]);
        }
        // ---- stage 1: dY = dH1, X = x0
#pragma unroll
        for (int kb = 0; kb < W::NK; ++kb) out[(W::S1Y + kb) * 64] = dh1[kb];
        {                                                       // dX0 = W1^T dH1 -> grad_feats
            const floatx16 acc = back_block_w<W::NK>(w1t, dh1);
            if (live) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (in_dim != IN)
                        store_grad4_narrow<TIO>(grad_feats + s * in_dim, 8 * q + 4 * g, in_dim, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                    else
                        store_grad4<TIO>(grad_feats + s * IN + 8 * q + 4 * g, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward: dW kernel
// Workgroup = 8 waves, 4 tiles per round.  Waves 0..3 re-derive the X operands of "their" tile (forward pass, 68 MFMAs) and
// write the X images; waves 4..7 fetch the dY operands of the same four tiles from the scratch (all 26 KB of a tile in flight
// at once) and write the dY images (the layouts store_chained / store_natural / load_transposed define).  After a barrier ALL
// eight waves add the products of THEIR dW blocks for the four tiles: wave w owns dY block it = w of every HH-wide dY (and
// its bias sums), and X block kt = w of the two 16-wide dY's (dY5, dY2; their bias sums go to wave 0) - 15 blocks of 16x16
// (60 VGPRs) per wave, two waves per SIMD.
constexpr int WD_TILES = 4;
constexpr int WD_WAVES = 2 * WD_TILES;

template <int HH> struct GradW {
    static constexpr int NK = Wide<HH>::NK;
    static_assert(NK == WD_WAVES, "block ownership below is written for hidden = 16 x (number of waves) = 128");
    floatx4 dW5, dW4[NK], dW3[3], dW2, dW1[2];
    floatx4 db;          // rows: 0 layer-4 block `wave`, 1 layer-1 block `wave`, 2 layer 2 (wave 0), 3 layer 5 (wave 0)
};

template <int HH, typename TIO>
__global__ void __launch_bounds__(WD_WAVES * 64)
wide_dw_kernel(const TIO* __restrict__ feats, const float* __restrict__ dirs, int64_t num_samples, int in_dim,
               const float* __restrict__ params, const bf16x8* __restrict__ scratch, int accumulate, float* __restrict__ partials) {
    typedef Wide<HH> W;
    constexpr int NK = W::NK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    // forward weight images + bias vectors (for the X operands), then [4 tiles][Y image | X image]
    __bf16* sw = reinterpret_cast<__bf16*>(smem_all);
    float* biasv = reinterpret_cast<float*>(smem_all + (size_t)W::L_FWD_END * 2);
    unsigned char* smem = smem_all + (size_t)W::L_FWD_END * 2 + (size_t)W::BIASV_FLOATS * 4;
    stage_weights_wide<HH, false>(sw, biasv, params, in_dim, threadIdx.x, WD_WAVES * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LaneW<HH> L;
    lane_const_wide<HH>(L, sw, biasv, params, in_dim, lane);
    __syncthreads();
    const int64_t num_tiles = (num_samples + TS - 1) / TS;
    const int n = lane & 31, g = lane >> 5;
    const int wc_off = (2 * n + g) * 8, wn_off = g * TILE_REGION + n * 16;
    const int tr_off = ((lane >> 1) & 1) * TILE_REGION + (8 * (lane >> 4) + 2 * ((lane >> 2) & 3) + (lane & 1)) * 8;
    const bool x_role = wave < WD_TILES;                       // waves 0..3: X operands; waves 4..7: dY operands
    const int my_tile = wave & (WD_TILES - 1);
    unsigned char* imgY = smem + (size_t)my_tile * 2 * W::IMG_BYTES;
    unsigned char* imgX = imgY + W::IMG_BYTES;
    GradW<HH> G;
    G.dW5 = zero4(); G.dW2 = zero4(); G.db = zero4();
#pragma unroll
    for (int i = 0; i < NK; ++i) G.dW4[i] = zero4();
#pragma unroll
    for (int i = 0; i < 3; ++i) G.dW3[i] = zero4();
    G.dW1[0] = zero4(); G.dW1[1] = zero4();
    const int64_t rounds = (num_tiles + WD_TILES - 1) / WD_TILES;
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, (u32x4)(0u));
    for (int64_t rd = blockIdx.x; rd < rounds; rd += gridDim.x) {
        const int64_t tile = rd * WD_TILES + my_tile;
        const bool have = tile < num_tiles;
        const bf16x8* in = scratch + tile * (int64_t)(W::NSLOT * 64) + lane;
#define LOADK(slot) ((have && !x_role) ? in[(slot) * 64] : zero8)
        // bias sum of the dY block held in operand `a` into row `row` of the shared block
#define BIAS_ROW(a, row) { const unsigned one2 = (lane & 15) == (row) ? 0x3f803f80u : 0u; const u32x4 ones = {one2, one2, one2, one2}; \
                           G.db = mma16(__builtin_bit_cast(bf16x8, ones), (a), G.db); }
        // one register set for both roles (a wave has one role): in the dY role the fetched operands live in the fields of A
        ActsW<HH> A;
        bf16x8& r5 = A.x0[0];
        bf16x8& r2 = A.x0[1];
        bf16x8 (&r4)[NK] = A.h2;
        bf16x8 (&r3)[NK] = A.h3;
        bf16x8 (&r1)[NK] = A.h1;
        if (x_role) {                                          // wave-uniform branch
            float d[3];
            const int64_t sidx = tile * TS + n;
            fetch_inputs_wide<TIO>(feats, dirs, sidx, have && sidx < num_samples, g, in_dim, A.x0, d);
            forward_tile_wide<HH>(L, d, A);
        } else {                                               // all 26 operand blocks of the tile in flight at once
            r5 = LOADK(W::S5Y);
            r2 = LOADK(W::S2Y);
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) r4[kb] = LOADK(W::S4Y + kb);
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) r3[kb] = LOADK(W::S3Y + kb);
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) r1[kb] = LOADK(W::S1Y + kb);
        }
        // ---------------- stage 5: dY5 natural [1 kb] x h3 chained [NK kb] -> dW5 (this wave: X block kt = wave)
        if (x_role) {
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) store_chained(imgX + wc_off, kb, A.h3[kb]);
        } else {
            store_natural(imgY + wn_off, 0, r5);
        }
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a = load_transposed(img, 0);
            if (wave == 0) BIAS_ROW(a, 3)
            G.dW5 = mma16(a, load_transposed(img + W::IMG_BYTES, wave), G.dW5);
        }
        __syncthreads();
        // ---------------- stage 4: dH3 [NK] x h2 [NK] -> dW4 (this wave: dY block it = wave), b4
        if (x_role) {
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) store_chained(imgX + wc_off, kb, A.h2[kb]);
        } else {
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) store_chained(imgY + wc_off, kb, r4[kb]);
        }
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a = load_transposed(img, wave);
            BIAS_ROW(a, 0)
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) G.dW4[kt] = mma16(a, load_transposed(img + W::IMG_BYTES, kt), G.dW4[kt]);
        }
        __syncthreads();
        // ---------------- stage 3: dH2 [NK] x x2 [chained block 0, natural blocks 1, 2] -> dW3 (+ b3 on the ones slot)
        if (x_role) {
            store_chained(imgX + wc_off, 0, A.x2[0]);
            store_natural(imgX + wn_off, 1, A.x2[1]);
            store_natural(imgX + wn_off, 2, A.x2[2]);
        } else {
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) store_chained(imgY + wc_off, kb, r3[kb]);
        }
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a = load_transposed(img, wave);
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) G.dW3[kt] = mma16(a, load_transposed(img + W::IMG_BYTES, kt), G.dW3[kt]);
        }
        __syncthreads();
        // ---------------- stage 2: dY2 chained [1] x h1 [NK] -> dW2 (X block kt = wave), b2 (wave 0)
        if (x_role) {
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) store_chained(imgX + wc_off, kb, A.h1[kb]);
        } else {
            store_chained(imgY + wc_off, 0, r2);
        }
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a = load_transposed(img, 0);
            if (wave == 0) BIAS_ROW(a, 2)
            G.dW2 = mma16(a, load_transposed(img + W::IMG_BYTES, wave), G.dW2);
        }
        __syncthreads();
        // ---------------- stage 1: dH1 [NK] x x0 natural [2] -> dW1, b1
        if (x_role) {
            store_natural(imgX + wn_off, 0, A.x0[0]);
            store_natural(imgX + wn_off, 1, A.x0[1]);
        } else {
#pragma unroll
            for (int kb = 0; kb < NK; ++kb) store_chained(imgY + wc_off, kb, r1[kb]);
        }
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a = load_transposed(img, wave);
            BIAS_ROW(a, 1)
            G.dW1[0] = mma16(a, load_transposed(img + W::IMG_BYTES, 0), G.dW1[0]);
            G.dW1[1] = mma16(a, load_transposed(img + W::IMG_BYTES, 1), G.dW1[1]);
        }
        __syncthreads();
#undef LOADK
#undef BIAS_ROW
    }
    // ---- the workgroup's partial row (canonical parameter order): every element has exactly one owner
    float* out = partials + (int64_t)blockIdx.x * W::NPARAM_PAD;
    const int col = lane & 15, rg = lane >> 4;
    auto put = [&](int idx, float v) { if (accumulate) out[idx] += v; else out[idx] = v; };
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = 4 * rg + rr;                                     // row inside a 16-row block
        if (row < 3) put(W::OW5 + row * HH + 16 * wave + col, G.dW5[rr]);       // stages 5 / 2: owned X block kt = wave
        put(W::OW2 + row * HH + 16 * wave + col, G.dW2[rr]);
        const int R = 16 * wave + row;                                   // stages 4 / 3 / 1: owned dY block -> output neuron R
#pragma unroll
        for (int k2 = 0; k2 < NK; ++k2) put(W::OW4 + R * HH + 16 * k2 + col, G.dW4[k2][rr]);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) put(W::OW1 + R * IN + 16 * k2 + col, G.dW1[k2][rr]);
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2) {
            const int u = 16 * k2 + col;                                 // feature inside the 48-wide colour input
            if (k2 == 0) { if (u >= 1) put(W::OW3 + R * X2 + u - 1, G.dW3[k2][rr]); }
            else if (u < ONES_SLOT) put(W::OW3 + R * X2 + u - 1, G.dW3[k2][rr]);
            else if (u == ONES_SLOT) put(W::OB3 + R, G.dW3[k2][rr]);
        }
        // shared bias block: row = combo, column = neuron inside its 16-block
        if (row == 0) put(W::OB4 + 16 * wave + col, G.db[rr]);
        else if (row == 1) put(W::OB1 + 16 * wave + col, G.db[rr]);
        else if (wave == 0 && row == 2) put(W::OB2 + col, G.db[rr]);
        else if (wave == 0 && row == 3 && col < 3) put(W::OB5 + col, G.db[rr]);
    }
}

// ---------------------------------------------------------------------------------------------- backward: dW kernel, pipelined
// Same job, same images, same scratch as wide_dw_kernel above, other division of labour (round 5).  There every round ran
// [four waves recompute the forward while four wait for their loads] -> 5 x [store images | barrier | all eight waves multiply,
// each LDS operand read right in front of the MFMA that needs it | barrier]: ~25 K cycles per 128 samples for ~4 K cycles of
// MFMA.  Here the two halves of that work overlap:
//   * producer waves 0..3 only recompute activations and store X images.  They run ONE ROUND AHEAD: while the consumers multiply
//     stage s of round r, a producer computes the next piece of round r + 1's forward pass (layer 1 under stage 5, layers 2 + 3
//     under stage 4, the two halves of layer 4 under stages 3 and 2; the colour output layer is not needed here and is skipped),
//     holding both rounds' activations in registers (<= ~160 of them live at once: a stage's activations die with its store);
//   * consumer waves 4..7 own ALL the dW blocks (wave c: dY blocks 2c, 2c + 1 of the 128-wide stages, X blocks 2c, 2c + 1 of the two
//     16-wide ones: 30 blocks = 120 VGPRs), fetch the dY operands of "their" tile one stage ahead (global loads issued in front of
//     a stage's MFMAs, stored to the Y image behind its second barrier) and read all LDS operands of a tile in one batch before
//     its MFMAs.
// One producer and one consumer per SIMD: the producer's 32x32x16 chain and the consumer's 16x16x32 products interleave on the
// matrix pipe.  Barriers per round: the same ten (images are single-buffered: 65 KB of weights + 80 KB of images).
// Numerics: identical products and identical per-block sums; a block's four tiles are added in the same order.
template <int HH> struct GradW2 {
    static constexpr int NK = Wide<HH>::NK;
    static_assert(NK == 8, "block ownership below is written for hidden = 128: two dY blocks per consumer wave");
    floatx4 dW5[2], dW4[2][NK], dW3[2][3], dW2[2], dW1[2][2];
    floatx4 db;      // rows: 0,1 layer-4 blocks 2c, 2c+1; 2,3 layer-1 blocks 2c, 2c+1; 4 layer 2 (c == 0); 5 layer 5 (c == 0)
};

// pieces of forward_tile_wide (same instructions, same order inside a layer)
template <int HH> DEV void fwd_l1(const LaneW<HH>& L, ActsW<HH>& A) {
    typedef Wide<HH> W;
    keep_lds_reads_here();
#pragma unroll
    for (int t = 0; t < W::NB; ++t) {
        floatx16 acc = *reinterpret_cast<const floatx16*>(L.bias_lds + t * 32);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc = mma32(lds_a(L.w1, t * 32 * W::LD1 + 16 * kb), A.x0[kb], acc);
        A.h1[2 * t] = pack8<0, true>(acc);
        A.h1[2 * t + 1] = pack8<8, true>(acc);
    }
}
template <int HH> DEV void fwd_l2_l3(const LaneW<HH>& L, const float d[3], ActsW<HH>& A) {
    typedef Wide<HH> W;
    keep_lds_reads_here();
    {
        floatx16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < W::NK; ++kb) acc = mma32(lds_a(L.w2, 16 * kb), A.h1[kb], acc);
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] += L.b2[r];
        A.y0 = acc[0];
        if (L.g == 0) acc[0] = 0.0f;
        A.x2[0] = pack8<0, false>(acc);
    }
    encode_dir(d, L.g, A.x2[1], A.x2[2]);
    keep_lds_reads_here();
#pragma unroll
    for (int t = 0; t < W::NB; ++t) {
        floatx16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) acc = mma32(lds_a(L.w3, t * 32 * W::LD3 + 16 * kb), A.x2[kb], acc);
        A.h2[2 * t] = pack8<0, true>(acc);
        A.h2[2 * t + 1] = pack8<8, true>(acc);
    }
}
template <int HH, int T0, int T1> DEV void fwd_l4(const LaneW<HH>& L, ActsW<HH>& A) {
    typedef Wide<HH> W;
#pragma unroll
    for (int t = T0; t < T1; ++t) {
        keep_lds_reads_here();
        floatx16 acc = *reinterpret_cast<const floatx16*>(L.bias_lds + 32 * W::NB + t * 32);
#pragma unroll
        for (int kb = 0; kb < W::NK; ++kb) acc = mma32(lds_a(L.w4, t * 32 * W::LD4 + 16 * kb), A.h2[kb], acc);
        A.h3[2 * t] = pack8<0, true>(acc);
        A.h3[2 * t + 1] = pack8<8, true>(acc);
    }
}

// The two roles are two separate loops (each with the same ten barriers per round) so that a wave's registers hold only its own
// role's state - one loop with role branches made the allocator keep both roles' state live (1 KB of spills per lane).
// The producers' and the consumers' barriers are DIFFERENT call sites that pair up by COUNT only: s_barrier counts arriving waves,
// not program counters, which is what the hardware does but nothing the HIP programming model promises.  Both loops therefore
// name every barrier by what it separates - WD_STAGE_BARRIER(stage, FILLED | DRAINED), stages 5 .. 1 - and
// tests/test_abi_and_host.py::test_wide_dw2_roles_issue_the_same_barrier_sequence holds the two sequences (and the prologue's
// single barrier on either side) to each other, so a barrier added to, dropped from or reordered in one role fails the CPU suite
// instead of hanging a GPU.  WISP_WIDE_DW=1 keeps the one-role, barrier-per-stage kernel; the GPU suite compares the two bitwise.
#define WD_FILLED 0                 // the stage's X / dY images are complete in LDS: consumers may read them
#define WD_DRAINED 1                // the consumers are done with the images: producers may overwrite them
#define FILLED WD_FILLED
#define DRAINED WD_DRAINED
#define WD_STAGE_BARRIER(stage, phase) do { static_assert((stage) >= 1 && (stage) <= 5 && ((phase) == WD_FILLED || (phase) == WD_DRAINED), \
                                                          "wide_dw2: unknown pipeline barrier"); __syncthreads(); } while (0)
template <int HH, typename TIO>
DEV void dw2_producer(const LaneW<HH>& L, const TIO* __restrict__ feats, const float* __restrict__ dirs, int64_t num_samples,
                      int in_dim, int64_t rounds, unsigned char* imgX, int my_tile, int lane) {
    typedef Wide<HH> W;
    constexpr int NK = W::NK;
    const int n = lane & 31, g = lane >> 5;
    const int wc_off = (2 * n + g) * 8, wn_off = g * TILE_REGION + n * 16;
    // A = this round's activations (complete), N = the next round's (computed piece by piece under the consumers' stages),
    // N.x0 / nd = the next round's inputs, fetched a round early
    ActsW<HH> A, N;
    float nd[3];
    const int64_t rd0 = blockIdx.x, step = gridDim.x;
    {
        float d[3];
        const int64_t s0 = (rd0 * WD_TILES + my_tile) * TS + n;
        fetch_inputs_wide<TIO>(feats, dirs, s0, rd0 < rounds && s0 < num_samples, g, in_dim, A.x0, d);
        fwd_l1<HH>(L, A);
        fwd_l2_l3<HH>(L, d, A);
        fwd_l4<HH, 0, W::NB>(L, A);
        const int64_t s1 = ((rd0 + step) * WD_TILES + my_tile) * TS + n;
        fetch_inputs_wide<TIO>(feats, dirs, s1, rd0 + step < rounds && s1 < num_samples, g, in_dim, N.x0, nd);
    }
    for (int64_t rd = rd0; rd < rounds; rd += step) {
        // stage 5: X = h3 (chained); meanwhile layer 1 of the next round
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) store_chained(imgX + wc_off, kb, A.h3[kb]);
        WD_STAGE_BARRIER(5, FILLED);
        fwd_l1<HH>(L, N);
        WD_STAGE_BARRIER(5, DRAINED);
        // stage 4: X = h2; meanwhile layers 2 + 3
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) store_chained(imgX + wc_off, kb, A.h2[kb]);
        WD_STAGE_BARRIER(4, FILLED);
        fwd_l2_l3<HH>(L, nd, N);
        WD_STAGE_BARRIER(4, DRAINED);
        // stage 3: X = x2 (chained block 0, natural blocks 1, 2); meanwhile the first half of layer 4
        store_chained(imgX + wc_off, 0, A.x2[0]);
        store_natural(imgX + wn_off, 1, A.x2[1]);
        store_natural(imgX + wn_off, 2, A.x2[2]);
        WD_STAGE_BARRIER(3, FILLED);
        fwd_l4<HH, 0, W::NB / 2>(L, N);
        WD_STAGE_BARRIER(3, DRAINED);
        // stage 2: X = h1; meanwhile the second half of layer 4
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) store_chained(imgX + wc_off, kb, A.h1[kb]);
        WD_STAGE_BARRIER(2, FILLED);
        fwd_l4<HH, W::NB / 2, W::NB>(L, N);
        WD_STAGE_BARRIER(2, DRAINED);
        // stage 1: X = x0 (natural, 2 blocks); meanwhile: hand over, fetch the inputs of the round after next
        store_natural(imgX + wn_off, 0, A.x0[0]);
        store_natural(imgX + wn_off, 1, A.x0[1]);
        WD_STAGE_BARRIER(1, FILLED);
        A = N;
        const int64_t r2 = rd + 2 * step;
        const int64_t s2 = (r2 * WD_TILES + my_tile) * TS + n;
        fetch_inputs_wide<TIO>(feats, dirs, s2, r2 < rounds && s2 < num_samples, g, in_dim, N.x0, nd);
        WD_STAGE_BARRIER(1, DRAINED);
    }
}

template <int HH>
DEV void dw2_consumer(const bf16x8* __restrict__ scratch, int64_t num_tiles, int64_t rounds, unsigned char* smem, int c, int lane,
                      int accumulate, float* __restrict__ partials) {
    typedef Wide<HH> W;
    constexpr int NK = W::NK;
    const int n = lane & 31, g = lane >> 5;
    const int wc_off = (2 * n + g) * 8, wn_off = g * TILE_REGION + n * 16;
    const int tr_off = ((lane >> 1) & 1) * TILE_REGION + (8 * (lane >> 4) + 2 * ((lane >> 2) & 3) + (lane & 1)) * 8;
    unsigned char* imgY = smem + (size_t)c * 2 * W::IMG_BYTES;
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, (u32x4)(0u));
    GradW2<HH> G;
    G.dW5[0] = G.dW5[1] = G.dW2[0] = G.dW2[1] = zero4(); G.db = zero4();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int k = 0; k < NK; ++k) G.dW4[i][k] = zero4();
#pragma unroll
        for (int k = 0; k < 3; ++k) G.dW3[i][k] = zero4();
        G.dW1[i][0] = zero4(); G.dW1[i][1] = zero4();
    }
    bf16x8 yb[NK];                                              // the dY blocks of the NEXT stage of this wave's tile
#define DY_PTR(rd_) (scratch + ((rd_) * WD_TILES + c) * (int64_t)(W::NSLOT * 64) + lane)
#define DY_HAVE(rd_) ((rd_) < rounds && (rd_) * WD_TILES + c < num_tiles)
#define BIAS_ROW(a, row) { const unsigned one2 = (lane & 15) == (row) ? 0x3f803f80u : 0u; const u32x4 ones = {one2, one2, one2, one2}; \
                           G.db = mma16(__builtin_bit_cast(bf16x8, ones), (a), G.db); }
    const int64_t rd0 = blockIdx.x, step = gridDim.x;
    yb[0] = DY_HAVE(rd0) ? DY_PTR(rd0)[W::S5Y * 64] : zero8;
    for (int64_t rd = rd0; rd < rounds; rd += step) {
        const bool have = DY_HAVE(rd);
        const bf16x8* in = DY_PTR(rd);
        // ================= stage 5: dY5 (natural, 1 block) x h3 -> dW5, b5
        store_natural(imgY + wn_off, 0, yb[0]);
        WD_STAGE_BARRIER(5, FILLED);
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) yb[kb] = have ? in[(W::S4Y + kb) * 64] : zero8;
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a = load_transposed(img, 0);
            const bf16x8 b0 = load_transposed(img + W::IMG_BYTES, 2 * c), b1 = load_transposed(img + W::IMG_BYTES, 2 * c + 1);
            if (c == 0) BIAS_ROW(a, 5)
            G.dW5[0] = mma16(a, b0, G.dW5[0]);
            G.dW5[1] = mma16(a, b1, G.dW5[1]);
        }
        WD_STAGE_BARRIER(5, DRAINED);
        // ================= stage 4: dH3 x h2 -> dW4, b4
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) store_chained(imgY + wc_off, kb, yb[kb]);
        WD_STAGE_BARRIER(4, FILLED);
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) yb[kb] = have ? in[(W::S3Y + kb) * 64] : zero8;
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a0 = load_transposed(img, 2 * c), a1 = load_transposed(img, 2 * c + 1);
            bf16x8 b[NK];
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) b[kt] = load_transposed(img + W::IMG_BYTES, kt);
            BIAS_ROW(a0, 0)
            BIAS_ROW(a1, 1)
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                G.dW4[0][kt] = mma16(a0, b[kt], G.dW4[0][kt]);
                G.dW4[1][kt] = mma16(a1, b[kt], G.dW4[1][kt]);
            }
        }
        WD_STAGE_BARRIER(4, DRAINED);
        // ================= stage 3: dH2 x x2 -> dW3 (+ b3 on the ones slot)
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) store_chained(imgY + wc_off, kb, yb[kb]);
        WD_STAGE_BARRIER(3, FILLED);
        yb[0] = have ? in[W::S2Y * 64] : zero8;
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a0 = load_transposed(img, 2 * c), a1 = load_transposed(img, 2 * c + 1);
            bf16x8 b[3];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) b[kt] = load_transposed(img + W::IMG_BYTES, kt);
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                G.dW3[0][kt] = mma16(a0, b[kt], G.dW3[0][kt]);
                G.dW3[1][kt] = mma16(a1, b[kt], G.dW3[1][kt]);
            }
        }
        WD_STAGE_BARRIER(3, DRAINED);
        // ================= stage 2: dY2 (chained, 1 block) x h1 -> dW2, b2
        store_chained(imgY + wc_off, 0, yb[0]);
        WD_STAGE_BARRIER(2, FILLED);
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) yb[kb] = have ? in[(W::S1Y + kb) * 64] : zero8;
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a = load_transposed(img, 0);
            const bf16x8 b0 = load_transposed(img + W::IMG_BYTES, 2 * c), b1 = load_transposed(img + W::IMG_BYTES, 2 * c + 1);
            if (c == 0) BIAS_ROW(a, 4)
            G.dW2[0] = mma16(a, b0, G.dW2[0]);
            G.dW2[1] = mma16(a, b1, G.dW2[1]);
        }
        WD_STAGE_BARRIER(2, DRAINED);
        // ================= stage 1: dH1 x x0 (natural, 2 blocks) -> dW1, b1
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) store_chained(imgY + wc_off, kb, yb[kb]);
        WD_STAGE_BARRIER(1, FILLED);
        yb[0] = DY_HAVE(rd + step) ? DY_PTR(rd + step)[W::S5Y * 64] : zero8;
#pragma unroll
        for (int tl = 0; tl < WD_TILES; ++tl) {
            const unsigned char* img = smem + (size_t)tl * 2 * W::IMG_BYTES + tr_off;
            const bf16x8 a0 = load_transposed(img, 2 * c), a1 = load_transposed(img, 2 * c + 1);
            const bf16x8 b0 = load_transposed(img + W::IMG_BYTES, 0), b1 = load_transposed(img + W::IMG_BYTES, 1);
            BIAS_ROW(a0, 2)
            BIAS_ROW(a1, 3)
            G.dW1[0][0] = mma16(a0, b0, G.dW1[0][0]);
            G.dW1[0][1] = mma16(a0, b1, G.dW1[0][1]);
            G.dW1[1][0] = mma16(a1, b0, G.dW1[1][0]);
            G.dW1[1][1] = mma16(a1, b1, G.dW1[1][1]);
        }
        WD_STAGE_BARRIER(1, DRAINED);
    }
#undef DY_PTR
#undef DY_HAVE
#undef BIAS_ROW
    // ---- the workgroup's partial row (canonical parameter order): every element has exactly one owner among the consumers
    float* out = partials + (int64_t)blockIdx.x * W::NPARAM_PAD;
    const int col = lane & 15, rg = lane >> 4;
    auto put = [&](int idx, float v) { if (accumulate) out[idx] += v; else out[idx] = v; };
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = 4 * rg + rr;                                     // row inside a 16-row block
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = 2 * c + i;
            if (row < 3) put(W::OW5 + row * HH + 16 * blk + col, G.dW5[i][rr]);       // stages 5 / 2: owned X blocks
            put(W::OW2 + row * HH + 16 * blk + col, G.dW2[i][rr]);
            const int R = 16 * blk + row;                                // stages 4 / 3 / 1: owned dY blocks -> output neuron R
#pragma unroll
            for (int k2 = 0; k2 < NK; ++k2) put(W::OW4 + R * HH + 16 * k2 + col, G.dW4[i][k2][rr]);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) put(W::OW1 + R * IN + 16 * k2 + col, G.dW1[i][k2][rr]);
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2) {
                const int u = 16 * k2 + col;                             // feature inside the 48-wide colour input
                if (k2 == 0) { if (u >= 1) put(W::OW3 + R * X2 + u - 1, G.dW3[i][k2][rr]); }
                else if (u < ONES_SLOT) put(W::OW3 + R * X2 + u - 1, G.dW3[i][k2][rr]);
                else if (u == ONES_SLOT) put(W::OB3 + R, G.dW3[i][k2][rr]);
            }
        }
        // shared bias block: row = combo, column = neuron inside its 16-block
        if (row == 0) put(W::OB4 + 16 * (2 * c) + col, G.db[rr]);
        else if (row == 1) put(W::OB4 + 16 * (2 * c + 1) + col, G.db[rr]);
        else if (row == 2) put(W::OB1 + 16 * (2 * c) + col, G.db[rr]);
        else if (row == 3) put(W::OB1 + 16 * (2 * c + 1) + col, G.db[rr]);
        else if (c == 0 && row == 4) put(W::OB2 + col, G.db[rr]);
        else if (c == 0 && row == 5 && col < 3) put(W::OB5 + col, G.db[rr]);
    }
}

template <int HH, typename TIO>
__global__ void __launch_bounds__(WD_WAVES * 64)
wide_dw2_kernel(const TIO* __restrict__ feats, const float* __restrict__ dirs, int64_t num_samples, int in_dim,
                const float* __restrict__ params, const bf16x8* __restrict__ scratch, int accumulate, float* __restrict__ partials) {
    typedef Wide<HH> W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    __bf16* sw = reinterpret_cast<__bf16*>(smem_all);
    float* biasv = reinterpret_cast<float*>(smem_all + (size_t)W::L_FWD_END * 2);
    unsigned char* smem = smem_all + (size_t)W::L_FWD_END * 2 + (size_t)W::BIASV_FLOATS * 4;      // [4 tiles][Y image | X image]
    stage_weights_wide<HH, false>(sw, biasv, params, in_dim, threadIdx.x, WD_WAVES * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t num_tiles = (num_samples + TS - 1) / TS;
    const int64_t rounds = (num_tiles + WD_TILES - 1) / WD_TILES;
    const int my_tile = wave & (WD_TILES - 1);
    if (wave < WD_TILES) {                                     // wave-uniform: whole waves take one side
        LaneW<HH> L;
        lane_const_wide<HH>(L, sw, biasv, params, in_dim, lane);
        __syncthreads();
        dw2_producer<HH, TIO>(L, feats, dirs, num_samples, in_dim, rounds, smem + (size_t)my_tile * 2 * W::IMG_BYTES + W::IMG_BYTES,
                              my_tile, lane);
    } else {
        __syncthreads();
        dw2_consumer<HH>(scratch, num_tiles, rounds, smem, my_tile, lane, accumulate, partials);
    }
}

// grad_params[packed(j)] += sum over the workgroups' partial rows
template <int HH>
__global__ void __launch_bounds__(1024)
wide_reduce_kernel(const float* __restrict__ partials, int rows, int in_dim, float* __restrict__ grad_params) {
    typedef Wide<HH> W;
    __shared__ float s[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx;
    float a = 0.0f;
    if (j < W::NPARAM)
        for (int r = ty; r < rows; r += 16) a += partials[(int64_t)r * W::NPARAM_PAD + j];
    s[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && j < W::NPARAM) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s[k][tx];
        const int dst = wide_packed_index(j, in_dim, HH);
        if (dst >= 0) grad_params[dst] += t;
    }
}

int cu_count_w() {
    static int n = [] { int d = 0, c = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d); return c > 0 ? c : 256; }();
    return n;
}

constexpr int64_t WIDE_CHUNK_SAMPLES = (int64_t)1 << 21;      // samples per chain + dW launch pair (scratch = 1.7 GB at hidden 128): every launch
                                                              // stages 125 KB of weight images per workgroup first, so fewer, longer launches

template <int HH, typename TIO>
int wide_forward(const void* feats, const float* dirs, int64_t S, int in_dim, const float* params, float* rgb, float* density,
                 hipStream_t st) {
    typedef Wide<HH> W;
    const size_t lds = (size_t)W::L_FWD_END * 2 + (size_t)W::BIASV_FLOATS * 4;
    auto kern = wide_fwd_kernel<HH, TIO>;
    const hipError_t e = WISP_ALLOW_LDS(kern, lds);
    if (e != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, "nerf_mlp_wide", hipGetErrorString(e));
    const int64_t ntiles = (S + TS - 1) / TS;
    const int grid = (int)min64(ceil_div64(ntiles, WF_WAVES), 2 * cu_count_w());      // 66 KB of LDS: two workgroups per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WF_WAVES * 64), lds, st, (const TIO*)feats, dirs, S, in_dim, params, rgb, density);
    return 0;
}

template <int HH, typename TIO>
int wide_backward(const void* feats, const float* dirs, int64_t S, int in_dim, const float* params, const float* grad_rgb,
                  const float* grad_density, void* grad_feats, float* grad_params, void* workspace, hipStream_t st) {
    typedef Wide<HH> W;
    const size_t lds_c = (size_t)W::L_BWD_END * 2 + (size_t)W::BIASV_FLOATS * 4;
    const size_t lds_d = (size_t)W::L_FWD_END * 2 + (size_t)W::BIASV_FLOATS * 4 + (size_t)WD_TILES * 2 * W::IMG_BYTES;
    auto kc = wide_chain_kernel<HH, TIO>;
    // WISP_WIDE_DW=1: the barrier-per-stage kernel of rounds 2-4; default: the producer / consumer pipeline
    static const bool old_dw = [] { const char* e = getenv("WISP_WIDE_DW"); return e && atoi(e) == 1; }();
    auto kd = old_dw ? wide_dw_kernel<HH, TIO> : wide_dw2_kernel<HH, TIO>;
    const hipError_t e1 = WISP_ALLOW_LDS(kc, lds_c);
    const hipError_t e2 = WISP_ALLOW_LDS(kd, lds_d);
    if (e1 != hipSuccess || e2 != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, "nerf_mlp_wide", "hipFuncSetAttribute");
    const int cus = cu_count_w();
    float* partials = (float*)workspace;                                             // [cus][NPARAM_PAD]
    bf16x8* scratch = (bf16x8*)((char*)workspace + (size_t)cus * W::NPARAM_PAD * 4);
    const size_t io = sizeof(TIO);
    int first = 1;
    for (int64_t s0 = 0; s0 < S; s0 += WIDE_CHUNK_SAMPLES) {
        const int64_t n = S - s0 < WIDE_CHUNK_SAMPLES ? S - s0 : WIDE_CHUNK_SAMPLES;
        const int64_t ntiles = (n + TS - 1) / TS;
        const int grid_c = (int)min64(ceil_div64(ntiles, WC_WAVES), cus);
        hipLaunchKernelGGL(kc, dim3(grid_c), dim3(WC_WAVES * 64), lds_c, st, (const TIO*)((const char*)feats + (size_t)s0 * in_dim * io),
                           dirs + s0 * 3, n, in_dim, params, grad_rgb + s0 * 3, grad_density + s0,
                           (TIO*)((char*)grad_feats + (size_t)s0 * in_dim * io), scratch);
        // always `cus` workgroups: the partial rows of ALL of them are summed, and a row must have been written once
        hipLaunchKernelGGL(kd, dim3(cus), dim3(WD_WAVES * 64), lds_d, st, (const TIO*)((const char*)feats + (size_t)s0 * in_dim * io),
                           dirs + s0 * 3, n, in_dim, params, scratch, first ? 0 : 1, partials);
        first = 0;
    }
    hipLaunchKernelGGL(wide_reduce_kernel<HH>, dim3((W::NPARAM + 63) / 64), dim3(1024), 0, st, partials, cus, in_dim, grad_params);
    return 0;
}

}  // namespace

namespace wisp_mlp {

bool wide_supported(int hidden) { return hidden == 128; }

int64_t wide_workspace_bytes(int64_t num_samples, int hidden) {
    const int64_t chunk = num_samples < WIDE_CHUNK_SAMPLES ? num_samples : WIDE_CHUNK_SAMPLES;
    const int64_t tiles = (chunk + TS - 1) / TS;
    const int nk = hidden / 16;
    const int64_t nslot = 2 + 3 * (int64_t)nk;
    const int64_t nparam = (int64_t)hidden * IN + hidden + 16 * hidden + 16 + (int64_t)hidden * X2 + hidden + (int64_t)hidden * hidden + hidden + 3 * hidden + 3;
    const int64_t npad = (nparam + 63) / 64 * 64;
    return (int64_t)cu_count_w() * npad * 4 + tiles * nslot * 1024 + 1024;
}

#define WIDE_DISPATCH(FN, ...)                                                                        \
    if (hidden != 128) return wisp_fail(WISP_ERR_UNSUPPORTED, "nerf_mlp_wide", "hidden must be 128");   \
    switch (dtype_io) {                                                                               \
        case WISP_F32: return FN<128, float>(__VA_ARGS__);                                            \
        case WISP_F16: return FN<128, __half>(__VA_ARGS__);                                           \
        default: return FN<128, __hip_bfloat16>(__VA_ARGS__);                                         \
    }

int wide_forward_dispatch(const void* feats, int dtype_io, const float* dirs, int64_t S, int in_dim, int hidden, const float* params,
                          float* rgb, float* density, hipStream_t st) {
    WIDE_DISPATCH(wide_forward, feats, dirs, S, in_dim, params, rgb, density, st)
}

int wide_backward_dispatch(const void* feats, int dtype_io, const float* dirs, int64_t S, int in_dim, int hidden, const float* params,
                           const float* grad_rgb, const float* grad_density, void* grad_feats, float* grad_params, void* workspace,
                           hipStream_t st) {
    WIDE_DISPATCH(wide_backward, feats, dirs, S, in_dim, params, grad_rgb, grad_density, grad_feats, grad_params, workspace, st)
}
#undef WIDE_DISPATCH

}  // namespace wisp_mlp
