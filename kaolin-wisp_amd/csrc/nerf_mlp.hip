// Fused radiance-field decoder for gfx950: density MLP -> geometry features + positional-encoded view
// direction -> colour MLP, forward and backward, on the matrix cores.
//
// Replaces NeuralRadianceField.rgba after grid.interpolate (wisp/models/nefs/nerf.py:245-264), i.e. two
// BasicDecoders (wisp/models/decoders/basic_decoders.py:73-101; 5 nn.Linear GEMMs with K <= 64 whose activations
// round-trip HBM in the reference), PositionalEmbedder.forward (wisp/models/embedders/positional_embedder.py:51-66)
// and the relu / sigmoid epilogues.
//
// This file holds the C entry points and the EXACT fp32 path (compute_dtype = f32, the 1e-4 parity contract); the bf16
// path that training runs on is nerf_mlp_bf16.hip (register-chained layers, different kernel design).
//
// fp32 kernel (persistent, one wave = one tile of 32 samples, no inter-wave sync in the main loop):
//   * all weights live in LDS for the whole launch (row-major [out][K], row stride + 1 so that the MFMA operand reads are
//     bank-conflict free);
//   * a layer is D[out][sample] = W[out][k] . X[sample][k] with v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain):
//     weights are the A operand, the wave's activation tile (LDS, [sample][k]) is the B operand, so each lane ends up
//     holding 16 output neurons of ONE sample (column = lane & 31).  Bias + relu are applied in registers and the result
//     is written back to LDS as the next layer's [sample][k] tile - activations never touch HBM;
//   * backward recomputes the forward into LDS, back-propagates dY through W^T with the same MFMA shape and
//     accumulates dW += dY^T X over the whole launch in 224 accumulator VGPRs per lane; per-wave partial dW go to a
//     workspace and a second tiny kernel reduces them, so no atomics are used at all.
// Fixed shape of this build: IN = 32 grid features, H = 64 hidden, 4 view frequencies (nerf_hash.yaml).
#include "wisp_common.h"
#include "nerf_mlp_shape.h"
#include <cstdlib>

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

using namespace wisp_mlp;
constexpr int K3 = 48;                    // X2 padded to the MFMA K granularity

template <typename TC> struct Traits;
template <> struct Traits<float> {
    static constexpr int PAD = 1;
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
// LDS geometry for compute type TC
template <typename TC> struct Geo {
    static constexpr int P = Traits<TC>::PAD;
    static constexpr int LDI = IN + P;      // stride of [.][IN] tiles
    static constexpr int LDH = H + P;       // stride of [.][H] (and [.][64]) tiles
    // weights (elements)
    static constexpr int W1 = 0, W2 = W1 + H * LDI, W3 = W2 + 32 * LDH, W4 = W3 + H * LDH, W5 = W4 + H * LDH,
                         WEND = W5 + 32 * LDH;
    // per-wave activation tiles (elements): x0 [32][LDI]; h1, x2, h2, h3, dya, dyb [32][LDH]
    static constexpr int A_X0 = 0, A_H1 = A_X0 + TS * LDI, A_X2 = A_H1 + TS * LDH, A_H2 = A_X2 + TS * LDH,
                         A_H3 = A_H2 + TS * LDH, A_DA = A_H3 + TS * LDH, A_DB = A_DA + TS * LDH, A_END = A_DB + TS * LDH;
    static constexpr int NBIAS = H + 32 + H + H + 32;     // b1, b2(pad 32), b3, b4, b5(pad 32) as float
    static constexpr int B1 = 0, B2 = H, B3 = H + 32, B4 = 2 * H + 32, B5 = 3 * H + 32;
};

template <typename TC>
constexpr size_t lds_bytes(int waves, bool bwd) {
    typedef Geo<TC> G;
    const int act = bwd ? G::A_END : G::A_DA;
    size_t b = (size_t)G::NBIAS * 4 + (size_t)G::WEND * sizeof(TC) + (size_t)waves * act * sizeof(TC);
    return (b + 15) / 16 * 16;
}

// ------------------------------------------------------------------------------------------------ MFMA helpers
// D[i][n] += sum_k A[i][k] * B[n][k]   (A = 32 weight rows, B = the wave's 32 samples), K multiple of the step
template <typename TC> struct MMA;

template <> struct MMA<float> {
    template <int K>
    static __device__ __forceinline__ void nt(const float* A, int lda, const float* B, int ldb, floatx16& acc, int lane) {
        const float* pa = A + (lane & 31) * lda + (lane >> 5);
        const float* pb = B + (lane & 31) * ldb + (lane >> 5);
#pragma unroll
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
    }
    // D[k][n] += sum_i W[i][krow0 + k] * dY[n][i]     (reduction over M rows of W)
    template <int M>
    static __device__ __forceinline__ void tn(const float* W, int ldw, int krow0, const float* dY, int ldy, floatx16& acc,
                                              int lane) {
        const float* pa = W + (lane >> 5) * ldw + krow0 + (lane & 31);
        const float* pb = dY + (lane & 31) * ldy + (lane >> 5);
#pragma unroll
        for (int i = 0; i < M; i += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[i * ldw], pb[i], acc, 0, 0, 0);
    }
};

// dW[i0 + r][k0 + c] += sum_n dY[n][i0 + r] * X[n][k0 + c]   (reduction over the wave's 32 samples).
// Exact fp32 MFMA, one sample pair per step; both operands are COLUMNS of the [sample][feature] LDS tiles.
// One whole layer: acc[it * KT + kt] += dY[:, it-th 32 columns]^T  X[:, kt-th 32 columns]; every operand column block
// is gathered from LDS once per sample step and reused by all the MFMAs that need it.
template <int IT, int KT>
static __device__ __forceinline__ void dw_layer(const float* dY, int ldy, const float* X, int ldx, floatx16* acc, int lane) {
    const float* pa = dY + (lane >> 5) * ldy + (lane & 31);
    const float* pb = X + (lane >> 5) * ldx + (lane & 31);
#pragma unroll
    for (int n = 0; n < TS; n += 2) {
        float a[IT], b[KT];
#pragma unroll
        for (int it = 0; it < IT; ++it) a[it] = pa[n * ldy + it * 32];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) b[kt] = pb[n * ldx + kt * 32];
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                acc[it * KT + kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[it], b[kt], acc[it * KT + kt], 0, 0, 0);
    }
}

// row of accumulator register `reg` for this lane (C/D layout of the 32x32 MFMA shapes)
static __device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

static __device__ __forceinline__ floatx16 zero16() {
    floatx16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

template <typename T> static __device__ __forceinline__ float io_to_f(T v);
template <> __device__ __forceinline__ float io_to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float io_to_f<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float io_to_f<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> static __device__ __forceinline__ T io_from_f(float v);
template <> __device__ __forceinline__ float io_from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half io_from_f<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ __hip_bfloat16 io_from_f<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

// ------------------------------------------------------------------------------------------------ the kernel
template <typename TC, typename TIO, int WAVES, bool BWD>
__global__ void __launch_bounds__(WAVES * 64, 1)
nerf_mlp_kernel(const TIO* __restrict__ feats, const float* __restrict__ dirs, int64_t num_samples, int in_dim,
                const float* __restrict__ params, float* __restrict__ out_rgb, float* __restrict__ out_density,
                const float* __restrict__ grad_rgb, const float* __restrict__ grad_density,
                TIO* __restrict__ grad_feats, float* __restrict__ partials) {
    typedef Geo<TC> G;
    typedef Traits<TC> Tr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sb = reinterpret_cast<float*>(smem);                                   // biases (fp32)
    TC* sw = reinterpret_cast<TC*>(smem + (size_t)G::NBIAS * 4);                  // weights
    constexpr int ACT = BWD ? G::A_END : G::A_DA;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    TC* act = sw + G::WEND + wave * ACT;
    const int n = lane & 31, half = lane >> 5;

    // ---- stage weights: zero everything (padding rows / columns must be 0), then copy the real entries
    for (int e = threadIdx.x; e < G::WEND; e += WAVES * 64) sw[e] = Tr::from_f(0.0f);
    for (int e = threadIdx.x; e < G::NBIAS; e += WAVES * 64) sb[e] = 0.0f;
    __syncthreads();
    for (int e = threadIdx.x; e < H * IN; e += WAVES * 64) sw[G::W1 + (e / IN) * G::LDI + e % IN] = Tr::from_f(packed_param(params, OW1 + e, in_dim));
    for (int e = threadIdx.x; e < 16 * H; e += WAVES * 64) sw[G::W2 + (e / H) * G::LDH + e % H] = Tr::from_f(packed_param(params, OW2 + e, in_dim));
    for (int e = threadIdx.x; e < H * X2; e += WAVES * 64) sw[G::W3 + (e / X2) * G::LDH + e % X2] = Tr::from_f(packed_param(params, OW3 + e, in_dim));
    for (int e = threadIdx.x; e < H * H; e += WAVES * 64) sw[G::W4 + (e / H) * G::LDH + e % H] = Tr::from_f(packed_param(params, OW4 + e, in_dim));
    for (int e = threadIdx.x; e < 3 * H; e += WAVES * 64) sw[G::W5 + (e / H) * G::LDH + e % H] = Tr::from_f(packed_param(params, OW5 + e, in_dim));
    for (int e = threadIdx.x; e < H; e += WAVES * 64) {
        sb[G::B1 + e] = packed_param(params, OB1 + e, in_dim); sb[G::B3 + e] = packed_param(params, OB3 + e, in_dim); sb[G::B4 + e] = packed_param(params, OB4 + e, in_dim);
    }
    if (threadIdx.x < 16) sb[G::B2 + threadIdx.x] = packed_param(params, OB2 + threadIdx.x, in_dim);
    if (threadIdx.x < 3) sb[G::B5 + threadIdx.x] = packed_param(params, OB5 + threadIdx.x, in_dim);
    __syncthreads();

    TC* x0 = act + G::A_X0; TC* h1 = act + G::A_H1; TC* x2 = act + G::A_X2; TC* h2 = act + G::A_H2; TC* h3 = act + G::A_H3;
    TC* dya = BWD ? act + G::A_DA : nullptr;
    TC* dyb = BWD ? act + G::A_DB : nullptr;

    // weight-gradient accumulators, live for the whole launch (BWD only)
    floatx16 dW1[2], dW2[2], dW3[4], dW4[4], dW5[2];
    float db1 = 0.f, db2 = 0.f, db3 = 0.f, db4 = 0.f, db5 = 0.f;     // lane c accumulates bias-gradient column c
    if (BWD) {
#pragma unroll
        for (int t = 0; t < 2; ++t) { dW1[t] = zero16(); dW2[t] = zero16(); dW5[t] = zero16(); }
#pragma unroll
        for (int t = 0; t < 4; ++t) { dW3[t] = zero16(); dW4[t] = zero16(); }
    }

    const int64_t ntiles = (num_samples + TS - 1) / TS;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t s = tile * TS + n;
        const bool live = s < num_samples;

        // ---- load the input tile: lane (n, half) copies 16 of the 32 features of sample n (vector loads)
        {
            TIO buf[16];
            if (live && in_dim == IN) {
                const uint4* src = reinterpret_cast<const uint4*>(feats + s * IN + half * 16);
                uint4* dstv = reinterpret_cast<uint4*>(buf);
#pragma unroll
                for (int q = 0; q < (int)(16 * sizeof(TIO) / 16); ++q) dstv[q] = src[q];
            } else if (live) {                      // narrow rows: in_dim elements, no alignment, zero padded
#pragma unroll
                for (int k = 0; k < 16; ++k) buf[k] = half * 16 + k < in_dim ? feats[s * in_dim + half * 16 + k] : io_from_f<TIO>(0.0f);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) x0[n * G::LDI + half * 16 + k] = Tr::from_f(live ? io_to_f<TIO>(buf[k]) : 0.0f);
        }
        // ---- positional encoding of the view direction -> x2[:, 15..41]; zero the K padding 42..63.
        // layout [d ; sin(2^k d) k-major ; cos(2^k d) k-major] (positional_embedder.py:61-65).  The two half-waves split the
        // work: half 0 writes d and the sines, half 1 the cosines and the padding; one accurate sinf / cosf per band
        // (bit-compatible with torch.sin / torch.cos to ~1 ulp).
        {
            float d[3] = {0.f, 0.f, 0.f};
            if (live) { d[0] = dirs[s * 3]; d[1] = dirs[s * 3 + 1]; d[2] = dirs[s * 3 + 2]; }
            TC* row = x2 + n * G::LDH + 15;
            if (half == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) row[a] = Tr::from_f(d[a]);
            } else {
#pragma unroll
                for (int k = X2; k < 64; ++k) x2[n * G::LDH + k] = Tr::from_f(0.0f);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    const float arg = (float)(1 << k) * d[a];
                    const float val = half == 0 ? sinf(arg) : cosf(arg);
                    row[3 + half * 3 * NF + k * 3 + a] = Tr::from_f(val);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- L1: h1 = relu(W1 x0 + b1)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx16 acc = zero16();
            MMA<TC>::template nt<IN>(sw + G::W1 + t * 32 * G::LDI, G::LDI, x0, G::LDI, acc, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = t * 32 + acc_row(r, lane);
                h1[n * G::LDH + row] = Tr::from_f(fmaxf(acc[r] + sb[G::B1 + row], 0.0f));
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- L2: y = W2 h1 + b2 ; density = relu(y0) ; geometry features y[1..15] -> x2[:, 0..14]
        float y0 = 0.0f;
        {
            floatx16 acc = zero16();
            MMA<TC>::template nt<H>(sw + G::W2, G::LDH, h1, G::LDH, acc, lane);
#pragma unroll
            for (int r = 0; r < 8; ++r) {                       // rows 0..15 live in regs 0..7 (rows 16.. are padding)
                const int row = acc_row(r, lane);
                const float y = acc[r] + sb[G::B2 + row];
                if (row == 0) y0 = y; else x2[n * G::LDH + row - 1] = Tr::from_f(y);
            }
            if (half == 0 && live && !BWD) out_density[s] = fmaxf(y0, 0.0f);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- L3: h2 = relu(W3 x2 + b3)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx16 acc = zero16();
            MMA<TC>::template nt<K3>(sw + G::W3 + t * 32 * G::LDH, G::LDH, x2, G::LDH, acc, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = t * 32 + acc_row(r, lane);
                h2[n * G::LDH + row] = Tr::from_f(fmaxf(acc[r] + sb[G::B3 + row], 0.0f));
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- L4: h3 = relu(W4 h2 + b4)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx16 acc = zero16();
            MMA<TC>::template nt<H>(sw + G::W4 + t * 32 * G::LDH, G::LDH, h2, G::LDH, acc, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = t * 32 + acc_row(r, lane);
                h3[n * G::LDH + row] = Tr::from_f(fmaxf(acc[r] + sb[G::B4 + row], 0.0f));
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- L5: rgb = sigmoid(W5 h3 + b5)   (rows 0..2 = regs 0..2 of the half-0 lanes)
        float sg[3] = {0.f, 0.f, 0.f};
        {
            floatx16 acc = zero16();
            MMA<TC>::template nt<H>(sw + G::W5, G::LDH, h3, G::LDH, acc, lane);
#pragma unroll
            for (int c = 0; c < 3; ++c) sg[c] = 1.0f / (1.0f + expf(-(acc[c] + sb[G::B5 + c])));
            if (half == 0 && live && !BWD) { out_rgb[s * 3] = sg[0]; out_rgb[s * 3 + 1] = sg[1]; out_rgb[s * 3 + 2] = sg[2]; }
        }
        if (!BWD) { __builtin_amdgcn_wave_barrier(); continue; }

        // ================================= backward =================================
        // dY5[n][c] = g_rgb * s (1 - s) in dya[:, 0..2]; columns 3..31 zero (they are read as MFMA padding)
        {
#pragma unroll
            for (int k = 0; k < 16; ++k) dya[n * G::LDH + half * 16 + k] = Tr::from_f(0.0f);
            __builtin_amdgcn_wave_barrier();
            if (half == 0 && live) {
#pragma unroll
                for (int c = 0; c < 3; ++c) dya[n * G::LDH + c] = Tr::from_f(grad_rgb[s * 3 + c] * sg[c] * (1.0f - sg[c]));
            }
            __builtin_amdgcn_wave_barrier();
        }
        // dW5 += dY5^T h3 ; db5 ; dH3 = (W5^T dY5) * (h3 > 0) -> dyb
        dw_layer<1, 2>(dya, G::LDH, h3, G::LDH, dW5, lane);
        if (lane < 3) { float a = 0.f; for (int m = 0; m < TS; ++m) a += Tr::to_f(dya[m * G::LDH + lane]); db5 += a; }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx16 acc = zero16();
            MMA<TC>::template tn<16>(sw + G::W5, G::LDH, t * 32, dya, G::LDH, acc, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = t * 32 + acc_row(r, lane);
                const float m = Tr::to_f(h3[n * G::LDH + row]) > 0.0f ? acc[r] : 0.0f;
                dyb[n * G::LDH + row] = Tr::from_f(m);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // dW4 += dH3^T h2 ; db4 ; dH2 = (W4^T dH3) * (h2 > 0) -> dya
        dw_layer<2, 2>(dyb, G::LDH, h2, G::LDH, dW4, lane);
        { float a = 0.f; for (int m = 0; m < TS; ++m) a += Tr::to_f(dyb[m * G::LDH + lane]); db4 += a; }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx16 acc = zero16();
            MMA<TC>::template tn<H>(sw + G::W4, G::LDH, t * 32, dyb, G::LDH, acc, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = t * 32 + acc_row(r, lane);
                const float m = Tr::to_f(h2[n * G::LDH + row]) > 0.0f ? acc[r] : 0.0f;
                dya[n * G::LDH + row] = Tr::from_f(m);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // dW3 += dH2^T x2 ; db3 ; dX2 = W3^T dH2 (only the 15 geometry columns carry gradient further)
        dw_layer<2, 2>(dya, G::LDH, x2, G::LDH, dW3, lane);
        { float a = 0.f; for (int m = 0; m < TS; ++m) a += Tr::to_f(dya[m * G::LDH + lane]); db3 += a; }
        {
            floatx16 acc = zero16();
            MMA<TC>::template tn<H>(sw + G::W3, G::LDH, 0, dya, G::LDH, acc, lane);
            // dY2[n][0] = g_density * (y0 > 0) ; dY2[n][c + 1] = dX2[c], c = 0..14 ; columns 16..31 zero
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; ++k) dyb[n * G::LDH + half * 16 + k] = Tr::from_f(0.0f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 8; ++r) {                       // dX2 rows 0..15 are in regs 0..7
                const int row = acc_row(r, lane);
                if (row < 15) dyb[n * G::LDH + row + 1] = Tr::from_f(acc[r]);
            }
            if (half == 0) dyb[n * G::LDH] = Tr::from_f((live && y0 > 0.0f) ? grad_density[s] : 0.0f);
        }
        __builtin_amdgcn_wave_barrier();
        // dW2 += dY2^T h1 ; db2 ; dH1 = (W2^T dY2) * (h1 > 0) -> dya
        dw_layer<1, 2>(dyb, G::LDH, h1, G::LDH, dW2, lane);
        if (lane < 16) { float a = 0.f; for (int m = 0; m < TS; ++m) a += Tr::to_f(dyb[m * G::LDH + lane]); db2 += a; }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx16 acc = zero16();
            MMA<TC>::template tn<16>(sw + G::W2, G::LDH, t * 32, dyb, G::LDH, acc, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = t * 32 + acc_row(r, lane);
                const float m = Tr::to_f(h1[n * G::LDH + row]) > 0.0f ? acc[r] : 0.0f;
                dya[n * G::LDH + row] = Tr::from_f(m);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // dW1 += dH1^T x0 ; db1 ; dX0 = W1^T dH1 -> grad_feats
        dw_layer<2, 1>(dya, G::LDH, x0, G::LDI, dW1, lane);
        { float a = 0.f; for (int m = 0; m < TS; ++m) a += Tr::to_f(dya[m * G::LDH + lane]); db1 += a; }
        {
            floatx16 acc = zero16();
            MMA<TC>::template tn<H>(sw + G::W1, G::LDI, 0, dya, G::LDH, acc, lane);
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc_row(r, lane) < in_dim) grad_feats[s * in_dim + acc_row(r, lane)] = io_from_f<TIO>(acc[r]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (BWD) {
        // per-wave partial gradients -> workspace row [block * WAVES + wave][NPARAM_PAD]; the reduce kernel sums rows.
        float* out = partials + ((int64_t)blockIdx.x * WAVES + wave) * NPARAM_PAD;
        const int col = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                out[OW1 + (it * 32 + row) * IN + col] = dW1[it][r];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    out[OW4 + (it * 32 + row) * H + kt * 32 + col] = dW4[it * 2 + kt][r];
                    if (kt * 32 + col < X2) out[OW3 + (it * 32 + row) * X2 + kt * 32 + col] = dW3[it * 2 + kt][r];
                }
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                if (row < 16) out[OW2 + row * H + kt * 32 + col] = dW2[kt][r];
                if (row < 3) out[OW5 + row * H + kt * 32 + col] = dW5[kt][r];
            }
        }
        out[OB1 + lane] = db1; out[OB3 + lane] = db3; out[OB4 + lane] = db4;
        if (lane < 16) out[OB2 + lane] = db2;
        if (lane < 3) out[OB5 + lane] = db5;
    }
}

// grad_params[packed(j)] += sum over partial rows (rows are in canonical order, nerf_mlp_shape.h).
// Block = 64 parameters x 16 row groups (row-strided partial sums, then LDS).
__global__ void __launch_bounds__(1024)
nerf_mlp_reduce_kernel(const float* __restrict__ partials, int rows, int in_dim, float* __restrict__ grad_params) {
    __shared__ float s[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx;
    float a = 0.0f;
    if (j < NPARAM)
        for (int r = ty; r < rows; r += 16) a += partials[(int64_t)r * NPARAM_PAD + j];
    s[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && j < NPARAM) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s[k][tx];
        const int dst = packed_index(j, in_dim);
        if (dst >= 0) grad_params[dst] += t;
    }
}

int cu_count() {
    static int n = [] { int d = 0, c = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d); return c > 0 ? c : 256; }();
    return n;
}

template <typename TC, typename TIO, int WAVES, bool BWD>
int launch(const void* feats, const float* dirs, int64_t s_total, int in_dim, const float* params, float* rgb, float* density,
           const float* grad_rgb, const float* grad_density, void* grad_feats, float* grad_params, float* workspace,
           hipStream_t st) {
    const size_t lds = lds_bytes<TC>(WAVES, BWD);
    auto kern = nerf_mlp_kernel<TC, TIO, WAVES, BWD>;
    const hipError_t e = WISP_ALLOW_LDS(kern, lds);
    if (e != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, "nerf_mlp", hipGetErrorString(e));
    const int64_t ntiles = (s_total + TS - 1) / TS;
    int grid = (int)min64(ceil_div64(ntiles, WAVES), cu_count());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, (const TIO*)feats, dirs, s_total, in_dim, params, rgb,
                       density, grad_rgb, grad_density, (TIO*)grad_feats, workspace);
    if (BWD)
        hipLaunchKernelGGL(nerf_mlp_reduce_kernel, dim3((NPARAM + 63) / 64), dim3(1024), 0, st, workspace, grid * WAVES,
                           in_dim, grad_params);
    return 0;
}

template <bool BWD>
int dispatch(const void* feats, int dtype_io, const float* dirs, int64_t s_total, int in_dim, const float* params, int compute,
             float* rgb, float* density, const float* grad_rgb, const float* grad_density, void* grad_feats,
             float* grad_params, float* workspace, hipStream_t st) {
#define WISP_MLP_GO(TC, W)                                                                                             \
    switch (dtype_io) {                                                                                                \
        case WISP_F32: return launch<TC, float, W, BWD>(feats, dirs, s_total, in_dim, params, rgb, density, grad_rgb, grad_density, grad_feats, grad_params, workspace, st); \
        case WISP_F16: return launch<TC, __half, W, BWD>(feats, dirs, s_total, in_dim, params, rgb, density, grad_rgb, grad_density, grad_feats, grad_params, workspace, st); \
        default: return launch<TC, __hip_bfloat16, W, BWD>(feats, dirs, s_total, in_dim, params, rgb, density, grad_rgb, grad_density, grad_feats, grad_params, workspace, st); \
    }
    if (compute == WISP_BF16) {                      // register-chained matrix-core kernels, nerf_mlp_bf16.hip
        if (!BWD) return wisp_mlp::bf16_forward(feats, dtype_io, dirs, s_total, in_dim, params, rgb, density, st);
        int rows = 0;
        if (int rc = wisp_mlp::bf16_backward(feats, dtype_io, dirs, s_total, in_dim, params, grad_rgb, grad_density, grad_feats,
                                             workspace, &rows, st))
            return rc;
        hipLaunchKernelGGL(nerf_mlp_reduce_kernel, dim3((NPARAM + 63) / 64), dim3(1024), 0, st, workspace, rows, in_dim, grad_params);
        return 0;
    }
    if (BWD) { WISP_MLP_GO(float, 1) }
    WISP_MLP_GO(float, 2)
#undef WISP_MLP_GO
}

}  // namespace

extern "C" int64_t wisp_nerf_mlp_param_count(int in_dim, int hidden, int view_freqs) {
    const int64_t pe = 3 + 6 * (int64_t)view_freqs;
    return (int64_t)hidden * in_dim + hidden + 16 * (int64_t)hidden + 16 + (int64_t)hidden * (15 + pe) + hidden +
           (int64_t)hidden * hidden + hidden + 3 * (int64_t)hidden + 3;
}

extern "C" int64_t wisp_nerf_mlp_workspace_floats(void) { return (int64_t)cu_count() * 4 * NPARAM_PAD; }

extern "C" int64_t wisp_nerf_mlp_bwd_workspace_bytes(int64_t num_samples, int hidden) {
    if (hidden == H) return wisp_nerf_mlp_workspace_floats() * 4;
    if (wisp_mlp::wide_supported(hidden) && num_samples >= 0) return wisp_mlp::wide_workspace_bytes(num_samples, hidden);
    return 0;
}

static int check_shape(int in_dim, int hidden, int view_freqs, int dtype_io, int compute) {
    if (in_dim < 1 || in_dim > IN || view_freqs != NF || !(hidden == H || wisp_mlp::wide_supported(hidden)))
        return wisp_fail(WISP_ERR_UNSUPPORTED, "nerf_mlp", "this build supports 1 <= in_dim <= 32, hidden 64 or 128, view_freqs=4");
    if (hidden != H && compute != WISP_BF16)
        return wisp_fail(WISP_ERR_UNSUPPORTED, "nerf_mlp", "hidden 128 is built for bf16 compute only (the exact fp32 path is hidden 64)");
    if (dtype_io != WISP_F32 && dtype_io != WISP_F16 && dtype_io != WISP_BF16) return wisp_fail(WISP_ERR_INVALID, "nerf_mlp", "bad dtype_io");
    if (compute != WISP_F32 && compute != WISP_BF16) return wisp_fail(WISP_ERR_INVALID, "nerf_mlp", "compute must be f32 or bf16");
    return 0;
}

extern "C" int wisp_nerf_mlp_fwd(const void* feats, int dtype_io, const float* dirs, int64_t num_samples, int in_dim,
                                 int hidden, int view_freqs, const float* params, int compute_dtype, float* rgb,
                                 float* density, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0, "negative count");
    if (int rc = check_shape(in_dim, hidden, view_freqs, dtype_io, compute_dtype)) return rc;
    if (num_samples == 0) return WISP_OK;
    WISP_REQUIRE(feats && dirs && params && rgb && density, "null pointer");
    if (hidden != H) {
        if (int rc = wisp_mlp::wide_forward_dispatch(feats, dtype_io, dirs, num_samples, in_dim, hidden, params, rgb, density,
                                                     (hipStream_t)stream))
            return rc;
        WISP_CHECK_LAUNCH();
        return WISP_OK;
    }
    if (int rc = dispatch<false>(feats, dtype_io, dirs, num_samples, in_dim, params, compute_dtype, rgb, density, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, (hipStream_t)stream))
        return rc;
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_nerf_mlp_bwd(const void* feats, int dtype_io, const float* dirs, int64_t num_samples, int in_dim,
                                 int hidden, int view_freqs, const float* params, int compute_dtype, const float* grad_rgb,
                                 const float* grad_density, void* grad_feats, float* grad_params, float* workspace,
                                 int64_t workspace_bytes, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0, "negative count");
    if (int rc = check_shape(in_dim, hidden, view_freqs, dtype_io, compute_dtype)) return rc;
    if (num_samples == 0) return WISP_OK;
    WISP_REQUIRE(feats && dirs && params && grad_rgb && grad_density && grad_feats && grad_params && workspace, "null pointer");
    WISP_REQUIRE(workspace_bytes >= wisp_nerf_mlp_bwd_workspace_bytes(num_samples, hidden), "workspace too small (wisp_nerf_mlp_bwd_workspace_bytes)");
    if (hidden != H) {
        if (int rc = wisp_mlp::wide_backward_dispatch(feats, dtype_io, dirs, num_samples, in_dim, hidden, params, grad_rgb, grad_density,
                                                      grad_feats, grad_params, workspace, (hipStream_t)stream))
            return rc;
        WISP_CHECK_LAUNCH();
        return WISP_OK;
    }
    if (int rc = dispatch<true>(feats, dtype_io, dirs, num_samples, in_dim, params, compute_dtype, nullptr, nullptr, grad_rgb,
                                grad_density, grad_feats, grad_params, workspace, (hipStream_t)stream))
        return rc;
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---- view direction per RAY (hidden 64, bf16 compute, 32-wide 16-bit feature rows: the training shape) ----------------
// The reference gathers ray directions per sample (packed_rf_tracer.py: rays.dirs.index_select(0, ridx)) and encodes every
// sample; here the encoding is done once per ray and the decoder kernels gather the 64-byte code by ray index.
extern "C" int wisp_nerf_mlp_dir_code(const float* ray_dirs, int64_t num_rays, int view_freqs, void* code, wisp_stream_t stream) {
    WISP_REQUIRE(num_rays >= 0, "negative count");
    WISP_REQUIRE(view_freqs == NF, "this build supports view_freqs=4");
    if (num_rays == 0) return WISP_OK;
    WISP_REQUIRE(ray_dirs && code, "null pointer");
    wisp_mlp::bf16_dir_code(ray_dirs, num_rays, code, (hipStream_t)stream);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

static int check_rays_shape(int in_dim, int hidden, int view_freqs, int dtype_io) {
    if (hidden != H || view_freqs != NF || !wisp_mlp::bf16_rays_supported(dtype_io, in_dim))
        return wisp_fail(WISP_ERR_UNSUPPORTED, "nerf_mlp_rays", "per-ray view codes: 1 <= in_dim <= 32, hidden 64, view_freqs 4, f32 / f16 / bf16 features");
    return 0;
}

extern "C" int wisp_nerf_mlp_fwd_rays(const void* feats, int dtype_io, const void* dir_code, const int64_t* ridx,
                                      int64_t num_samples, int in_dim, int hidden, int view_freqs, const float* params,
                                      float* rgb, float* density, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0, "negative count");
    if (int rc = check_rays_shape(in_dim, hidden, view_freqs, dtype_io)) return rc;
    if (num_samples == 0) return WISP_OK;
    WISP_REQUIRE(feats && dir_code && ridx && params && rgb && density, "null pointer");
    if (int rc = wisp_mlp::bf16_forward_rays(feats, dtype_io, dir_code, ridx, num_samples, in_dim, params, rgb, density, (hipStream_t)stream))
        return rc;
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_nerf_mlp_bwd_rays(const void* feats, int dtype_io, const void* dir_code, const int64_t* ridx,
                                      int64_t num_samples, int in_dim, int hidden, int view_freqs, const float* params,
                                      const float* grad_rgb, const float* grad_density, void* grad_feats, float* grad_params,
                                      float* workspace, int64_t workspace_bytes, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0, "negative count");
    if (int rc = check_rays_shape(in_dim, hidden, view_freqs, dtype_io)) return rc;
    if (num_samples == 0) return WISP_OK;
    WISP_REQUIRE(feats && dir_code && ridx && params && grad_rgb && grad_density && grad_feats && grad_params && workspace, "null pointer");
    WISP_REQUIRE(workspace_bytes >= wisp_nerf_mlp_bwd_workspace_bytes(num_samples, hidden), "workspace too small (wisp_nerf_mlp_bwd_workspace_bytes)");
    int rows = 0;
    if (int rc = wisp_mlp::bf16_backward_rays(feats, dtype_io, dir_code, ridx, num_samples, in_dim, params, grad_rgb, grad_density, grad_feats,
                                              workspace, &rows, (hipStream_t)stream))
        return rc;
    hipLaunchKernelGGL(nerf_mlp_reduce_kernel, dim3((NPARAM + 63) / 64), dim3(1024), 0, (hipStream_t)stream, workspace, rows, in_dim, grad_params);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}
