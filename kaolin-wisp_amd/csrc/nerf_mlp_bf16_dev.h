// Device helpers shared by the register-chained bf16 decoder kernels (nerf_mlp_bf16.hip: hidden 64, tuned;
// nerf_mlp_wide.hip: any hidden width that is a multiple of 32): vector types, the chained-K permutation, packing /
// masking of accumulator registers, MFMA wrappers, the LDS transposition images and the view-direction encoding.
#pragma once
#include "wisp_common.h"
#include "nerf_mlp_shape.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define DEV static __device__ __forceinline__

namespace wisp_mlp_dev {
using namespace wisp_mlp;

// ---------------------------------------------------------------------------------------------- LDS image of the weights
// slot p = 8 g + j of a chained 16-block <-> neuron offset
__host__ __device__ constexpr int phi16(int p) { return 8 * ((p & 7) >> 2) + 4 * (p >> 3) + (p & 3); }
__host__ __device__ constexpr int phi(int s) { return (s & ~15) + phi16(s & 15); }

constexpr int TILE_REGION = 640;            // bytes between feature-octet regions of a transposition image
constexpr int ONES_SLOT = 16 + PE;          // = 43: feature index (within the 48-wide colour input) that holds 1.0

// ---------------------------------------------------------------------------------------------- register helpers
DEV int acc_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

DEV floatx16 zero16() { floatx16 z; _Pragma("unroll") for (int r = 0; r < 16; ++r) z[r] = 0.0f; return z; }
DEV floatx4 zero4() { floatx4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// two floats -> one dword of two bf16 (v_cvt_pk_bf16_f32); the explicit pair keeps the compiler from converting singly
DEV unsigned cvt2(float a, float b) {
    const floatx2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
// NOTE: every 16-bit lane operation below is written on the WHOLE 8-element vector.  Bit-casting one extracted dword to
// a 2 x i16 vector (u32x4 -> [j] -> s16x2) is mis-folded by this compiler: all four dwords end up using dword 0.
template <int BASE, bool RELU> DEV bf16x8 pack8(const floatx16& a) {
    u32x4 w;
    _Pragma("unroll") for (int j = 0; j < 4; ++j) w[j] = cvt2(a[BASE + 2 * j], a[BASE + 2 * j + 1]);
    if (RELU) {                                  // sign bit set <=> negative: integer max with 0 on the bit patterns
        const s16x8 s = __builtin_elementwise_max(__builtin_bit_cast(s16x8, w), (s16x8)(0));
        return __builtin_bit_cast(bf16x8, s);
    }
    return __builtin_bit_cast(bf16x8, w);
}
DEV bf16x8 pack8f(const float v[8]) {
    u32x4 w;
    _Pragma("unroll") for (int j = 0; j < 4; ++j) w[j] = cvt2(v[2 * j], v[2 * j + 1]);
    return __builtin_bit_cast(bf16x8, w);
}
// BASE.. of `a` as bf16 where the relu output h is positive (bit pattern 0 or positive), else 0
template <int BASE> DEV bf16x8 pack8_masked(const floatx16& a, bf16x8 h) {
    // per 16-bit lane: d * min(h, 1) (h is 0 or a positive bit pattern), i.e. d where h > 0 else 0.  Spelled as two packed
    // instructions: left to itself the compiler turns any such expression into compares, selects and a permute per dword.
    const u32x4 hw = __builtin_bit_cast(u32x4, h);
    u32x4 w;
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {
        const unsigned d = cvt2(a[BASE + 2 * j], a[BASE + 2 * j + 1]);
        asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_mul_lo_u16 %0, %0, %2" : "=&v"(w[j]) : "v"(hw[j]), "v"(d));
    }
    return __builtin_bit_cast(bf16x8, w);
}
DEV floatx16 mma32(bf16x8 a, bf16x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEV floatx4 mma16(bf16x8 a, bf16x8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// A operand: 8 consecutive slots of one LDS row
DEV bf16x8 lds_a(const __bf16* lane_row, int elem_off) { return *reinterpret_cast<const bf16x8*>(lane_row + elem_off); }

// transposition image: per-lane write pointers  wc = img + (2 n + g) * 8  (chained halves), wn = img + g * 640 + n * 16
DEV void store_chained(unsigned char* wc, int kb, bf16x8 p) {
    bf16x4 lo = {p[0], p[1], p[2], p[3]}, hi = {p[4], p[5], p[6], p[7]};
    *reinterpret_cast<bf16x4*>(wc + (2 * kb) * TILE_REGION) = lo;
    *reinterpret_cast<bf16x4*>(wc + (2 * kb + 1) * TILE_REGION) = hi;
}
DEV void store_natural(unsigned char* wn, int kb, bf16x8 p) { *reinterpret_cast<bf16x8*>(wn + 2 * kb * TILE_REGION) = p; }
// operand of the 16x16x32 MFMA for feature block fb: lane (feature l & 15, kg = l >> 4) gets samples 4 kg + {0..3} and
// 16 + 4 kg + {0..3};  tr = img + ((l >> 1) & 1) * 640 + (8 kg + 2 ((l >> 2) & 3) + (l & 1)) * 8
DEV bf16x8 load_transposed(const unsigned char* tr, int fb) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tr + fb * 2 * TILE_REGION));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tr + fb * 2 * TILE_REGION + 256));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <typename T> DEV float io_to_f(T v);
template <> __device__ __forceinline__ float io_to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float io_to_f<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float io_to_f<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

// 8 consecutive input features as a bf16 operand
template <typename TIO> DEV bf16x8 load_feats8(const TIO* p, bool live) {
    bf16x8 v;
    if (!live) { _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f; return v; }
    if (sizeof(TIO) == 2 && !__is_same(TIO, __half)) return *reinterpret_cast<const bf16x8*>(p);
    _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] = (__bf16)io_to_f<TIO>(p[j]);
    return v;
}
// in_dim < 32: rows are in_dim elements long and carry no alignment; columns >= in_dim read as zero / are not written
template <typename TIO> DEV bf16x8 load_feats8_narrow(const TIO* row, int col0, int in_dim, bool live) {
    bf16x8 v;
    _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] = (__bf16)((live && col0 + j < in_dim) ? io_to_f<TIO>(row[col0 + j]) : 0.0f);
    return v;
}
template <typename T> DEV T io_from_f(float v);
template <> __device__ __forceinline__ float io_from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half io_from_f<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ __hip_bfloat16 io_from_f<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <typename TIO> DEV void store_grad4_narrow(TIO* row, int col0, int in_dim, float a, float b, float c, float d) {
    const float v[4] = {a, b, c, d};
    _Pragma("unroll") for (int j = 0; j < 4; ++j) if (col0 + j < in_dim) row[col0 + j] = io_from_f<TIO>(v[j]);
}
template <typename TIO> DEV void store_grad4(TIO* p, float a, float b, float c, float d) {
    if (sizeof(TIO) == 2 && !__is_same(TIO, __half)) {
        bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
        *reinterpret_cast<bf16x4*>(p) = v;
    } else if (sizeof(TIO) == 2) {
        __half2* q = reinterpret_cast<__half2*>(p);
        q[0] = __floats2half2_rn(a, b); q[1] = __floats2half2_rn(c, d);
    } else {
        float4 v = {a, b, c, d};
        *reinterpret_cast<float4*>(p) = v;
    }
}

// view-direction encoding [d ; sin(2^k d) k-major ; cos(2^k d) k-major] (positional_embedder.py:61-65) + the ones slot, as the
// two natural-order K blocks of the colour input: element e = 16 (kb - 1) + 8 g + j
DEV void encode_dir(const float d[3], int g, bf16x8& k1, bf16x8& k2) {
    float S[NF][3], C[NF][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float sv, cv;
        __sincosf(d[a], &sv, &cv);
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            S[k][a] = sv; C[k][a] = cv;
            const float s2 = 2.0f * sv * cv, c2 = 1.0f - 2.0f * sv * sv;      // angle doubling
            sv = s2; cv = c2;
        }
    }
    const float a1[8] = {d[0], d[1], d[2], S[0][0], S[0][1], S[0][2], S[1][0], S[1][1]};
    const float b1[8] = {S[1][2], S[2][0], S[2][1], S[2][2], S[3][0], S[3][1], S[3][2], C[0][0]};
    const float a2[8] = {C[0][1], C[0][2], C[1][0], C[1][1], C[1][2], C[2][0], C[2][1], C[2][2]};
    const float b2[8] = {C[3][0], C[3][1], C[3][2], 1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float v1[8], v2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v1[j] = g ? b1[j] : a1[j]; v2[j] = g ? b2[j] : a2[j]; }
    k1 = pack8f(v1); k2 = pack8f(v2);
}

}  // namespace wisp_mlp_dev
