// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/wisp_hip.h"

#define WISP_WAVE 64

extern thread_local char g_wisp_err[512];

static inline int wisp_fail(int code, const char* what, const char* detail) {
    snprintf(g_wisp_err, sizeof(g_wisp_err), "%s: %s", what, detail ? detail : "");
    return code;
}

#define WISP_REQUIRE(cond, what)                                              \
    do {                                                                      \
        if (!(cond)) return wisp_fail(WISP_ERR_INVALID, __func__, what);      \
    } while (0)

#define WISP_CHECK_LAUNCH()                                                   \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess)                                                 \
            return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e_)); \
    } while (0)

// Dynamic LDS beyond the default limit has to be allowed per kernel AND per device (one process may drive several devices):
// WISP_ALLOW_LDS(kernel, bytes) asks once per (call site = kernel instance, device) and remembers the grant.
static inline hipError_t wisp_allow_lds_once(const void* fn, size_t bytes, size_t* granted /* [64] */) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (bytes <= granted[dev]) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) granted[dev] = bytes;
    return e;
}
#define WISP_ALLOW_LDS(kern, bytes)                                                                     \
    ([&]() -> hipError_t {                                                                              \
        static size_t granted_[64] = {0};                                                               \
        return wisp_allow_lds_once(reinterpret_cast<const void*>(kern), (size_t)(bytes), granted_);      \
    })()

// torch.optim.AdamW's update of ONE element (decoupled decay, torch/optim/adamw.py _single_tensor_adamw), written once: the flat
// optimizer kernels (misc.hip) and the hash-grid reduce kernel's fused flush (hashgrid.hip) must produce the same bits.
// bc1 = 1 - beta1^step, bc2_sqrt = sqrt(1 - beta2^step); g already carries the caller's gradient scale.
static __device__ __forceinline__ void wisp_adamw_update(float& p, float& m, float& v, float g, float lr, float wd, float b1,
                                                         float b2, float eps, float bc1, float bc2_sqrt) {
    // Which products fuse into which sums is spelled out (and the compiler's own contraction switched off): left to itself it
    // fused b1 * m + (1 - b1) * g one way in one kernel and the other way in the next.
#pragma clang fp contract(off)
    p = p * (1.0f - lr * wd);
    m = __builtin_fmaf(b1, m, (1.0f - b1) * g);
    v = __builtin_fmaf(b2, v, ((1.0f - b2) * g) * g);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = __builtin_fmaf(-(lr / bc1), m / denom, p);
}
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }

// ---- storage-type conversion (tables / activations may be f32, f16 or bf16) ------------------------
template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Cvt<__half> {
    static __device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
    static __device__ __forceinline__ __half from_f(float v) { return __float2half_rn(v); }
};
template <> struct Cvt<__hip_bfloat16> {
    static __device__ __forceinline__ float to_f(__hip_bfloat16 v) { return __bfloat162float(v); }
    static __device__ __forceinline__ __hip_bfloat16 from_f(float v) { return __float2bfloat16(v); }
};

// Counter-based uniform [0,1) generator keyed by (seed, a, b); used only when the caller does not inject a jitter tensor.
// The (seed, a) part - a = the ray / nugget - goes through a 64-bit splitmix finaliser ONCE (in the marching kernels it is
// wave-uniform, i.e. scalar-unit work); the per-candidate part b is a 32-bit Weyl step + the lowbias32 finaliser: two
// 32-bit multiplies instead of the four 64-bit ones (quarter-rate v_mul_lo/hi_u32 chains) of a full 64-bit mix, which
// were ~40 % of the candidate loop.
static __host__ __device__ __forceinline__ uint32_t wisp_stream_key(uint64_t seed, uint64_t a) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (a + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z ^ (z >> 32));
}
static __host__ __device__ __forceinline__ float wisp_uniform01_keyed(uint32_t key, uint32_t b) {
    uint32_t x = key + 0x9E3779B9u * (b + 1u);
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f);    // 24 random bits -> [0,1)
}
static __host__ __device__ __forceinline__ float wisp_uniform01(uint64_t seed, uint64_t a, uint64_t b) {
    return wisp_uniform01_keyed(wisp_stream_key(seed, a), (uint32_t)b);
}

// Bit position of level-`level` cell (x, y, z) inside the occupancy bitfield (wisp_spc_build_bitfield): plain row-major
// (x fastest).  Three instructions per lookup; a Morton interleave cost ~9 per bit of `level` in the marching loop, and
// the whole field (256 KiB at level 7) is L2-resident either way.
static __host__ __device__ __forceinline__ uint32_t wisp_cell_bit(uint32_t x, uint32_t y, uint32_t z, int level) {
    return x | (y << level) | (z << (2 * level));
}

// CodebookOctreeGrid._index_features in TRAINING mode (codebook_grid.py:116-121) takes `y_soft.max(-1)[1]`: the argmax of the
// SOFTMAX VALUES, first index on ties (torch.max's documented rule) - not of the logits.  The two differ whenever an earlier logit
// lies so close below the maximum that exp(x - max) rounds to 1.0f (a gap under 2^-25: any exponential accurate to an ulp returns
// exactly 1 there), because then both entries carry the same softmax value and the EARLIER one wins.  With nerf_codebook.yaml's
// initialisation (logits ~ N(0, 0.01^2), 16 per row) that is ~1 row in 20 000 - found by the bench-shape parity test of round 6.
// logit_best = first index of the largest logit, mx = that logit, inv = 1 / sum exp(x - mx)  ->  the reference's index.
// (Evaluation mode indexes by torch.max(logits): logit_best itself.)
static __device__ __forceinline__ int codebook_softmax_pick(const float* __restrict__ row, int logit_best, float mx, float inv) {
    int best = logit_best;
    for (int k = logit_best - 1; k >= 0; --k) {
        const float d = row[k] - mx;
        if (d > -2.4e-7f && expf(d) * inv == inv) best = k;      // (the cheap bound first: 2^-22, anything below cannot round to 1)
    }
    return best;
}
