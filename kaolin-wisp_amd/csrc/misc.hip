// Library-level state (error string, ABI version) and the fused AdamW step.
//
// wisp_adamw_step replaces torch.optim.AdamW / apex FusedAdam as configured by BaseTrainer.init_optimizer
// (wisp/trainers/base_trainer.py:205-235, wisp/config/presets/torch.py:22-58): one pass over a flat fp32
// parameter buffer - 16 B read + 12 B written per parameter (+4 B when the gradient is zeroed in the same
// pass), against ~28 B/param/step for the unfused optimizer plus a separate zero_grad memset.
#include "wisp_common.h"

thread_local char g_wisp_err[512] = "";

extern "C" const char* wisp_last_error(void) { return g_wisp_err; }
extern "C" int wisp_abi_version(void) { return 1; }

__global__ void __launch_bounds__(256)
adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale, int zero_grad,
             __hip_bfloat16* __restrict__ shadow) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<float4*>(g + i);
            float4 mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gr = gp[k] * gscale;
                pp[k] *= (1.0f - lr * wd);                               // decoupled weight decay
                mp[k] = b1 * mp[k] + (1.0f - b1) * gr;
                vp[k] = b2 * vp[k] + (1.0f - b2) * gr * gr;
                const float denom = sqrtf(vp[k]) / bc2_sqrt + eps;
                pp[k] -= (lr / bc1) * (mp[k] / denom);
            }
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(v + i) = vv;
            if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (shadow) {                                                // bf16 copy of the updated parameters
#pragma unroll
                for (int k = 0; k < 4; ++k) shadow[i + k] = __float2bfloat16(pp[k]);
            }
        } else {
            for (int64_t j = i; j < n; ++j) {
                const float gr = g[j] * gscale;
                float pj = p[j] * (1.0f - lr * wd);
                const float mj = b1 * m[j] + (1.0f - b1) * gr;
                const float vj = b2 * v[j] + (1.0f - b2) * gr * gr;
                const float denom = sqrtf(vj) / bc2_sqrt + eps;
                pj -= (lr / bc1) * (mj / denom);
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (zero_grad) g[j] = 0.0f;
                if (shadow) shadow[j] = __float2bfloat16(pj);
            }
        }
    }
}

extern "C" int wisp_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                               int zero_grad, void* bf16_shadow, wisp_stream_t stream) {
    WISP_REQUIRE(n >= 0 && step >= 1, "bad n / step");
    if (n == 0) return WISP_OK;
    WISP_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null pointer");
    WISP_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) &&
                 ((uintptr_t)exp_avg_sq % 16 == 0), "buffers must be 16-byte aligned");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    const int64_t groups = ceil_div64(n, 4);
    const int grid = (int)min64(ceil_div64(groups, 256), 4096);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, const_cast<float*>(grad),
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale, zero_grad,
                       (__hip_bfloat16*)bf16_shadow);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}
