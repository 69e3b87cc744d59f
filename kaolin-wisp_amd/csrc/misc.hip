// Library-level state (error string, ABI version) and the fused AdamW step.
//
// wisp_adamw_step replaces torch.optim.AdamW / apex FusedAdam as configured by BaseTrainer.init_optimizer
// (wisp/trainers/base_trainer.py:205-235, wisp/config/presets/torch.py:22-58): one pass over a flat fp32
// parameter buffer - 16 B read + 12 B written per parameter (+4 B when the gradient is zeroed in the same
// pass), against ~28 B/param/step for the unfused optimizer plus a separate zero_grad memset.
#include "wisp_common.h"
#include <mutex>

thread_local char g_wisp_err[512] = "";

extern "C" const char* wisp_last_error(void) { return g_wisp_err; }
// 2: round-2 surface (workspace arguments, raytrace cache, *_rays, query, decode_rows, optimizer kinds)
// 3: hash-grid backward takes per-level slot scales (+ wisp_hashgrid_bwd_slot_stats), wisp_hashgrid_cells
extern "C" int wisp_abi_version(void) { return 4; }

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
                wisp_adamw_update(pp[k], mp[k], vp[k], gp[k] * gscale, lr, wd, b1, b2, eps, bc1, bc2_sqrt);
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
                float pj = p[j], mj = m[j], vj = v[j];
                wisp_adamw_update(pj, mj, vj, g[j] * gscale, lr, wd, b1, b2, eps, bc1, bc2_sqrt);
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

// ---- several parameter groups of ONE flat buffer in one launch (decoder | grid | rest, each with its own learning rate:
// base_trainer.py:216-235).  The decoder group is ~10 K parameters: as a launch of its own it costs more timeline than
// arithmetic.  Work is indexed by 4-float chunks across the groups; the arithmetic is adamw_kernel's.
#define ADAMW_MAX_GROUPS 4
struct AdamGroups {
    int n;
    int64_t begin[ADAMW_MAX_GROUPS], len[ADAMW_MAX_GROUPS], chunk0[ADAMW_MAX_GROUPS + 1];
    float lr[ADAMW_MAX_GROUPS], wd[ADAMW_MAX_GROUPS];
    __hip_bfloat16* shadow[ADAMW_MAX_GROUPS];
};

__global__ void __launch_bounds__(256)
adamw_groups_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                    AdamGroups gr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale, int zero_grad) {
    const int64_t total = gr.chunk0[gr.n];
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < ADAMW_MAX_GROUPS; ++q) k += (q < gr.n && c >= gr.chunk0[q]) ? 1 : 0;
        const int64_t off = (c - gr.chunk0[k]) * 4;                 // inside the group
        const int64_t i = gr.begin[k] + off;
        const int cnt = (int)(gr.len[k] - off < 4 ? gr.len[k] - off : 4);
        const float lr = gr.lr[k], wd = gr.wd[k];
        __hip_bfloat16* shadow = gr.shadow[k];
        float pv[4], gv[4], mv[4], vv[4];
        // (a group may start off a 16-byte boundary - the rows the hash-grid backward's fused update leaves over begin where a
        //  level begins, i.e. on a multiple of the feature width: such a group goes in 8-byte halves, anything else element
        //  by element)
        const bool vec = cnt == 4 && (i & 3) == 0;
        const bool vec2 = cnt == 4 && (i & 3) == 2;
        if (vec) {
            *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(p + i);
            *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(g + i);
            *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(m + i);
            *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(v + i);
        } else if (vec2) {
#pragma unroll
            for (int h = 0; h < 4; h += 2) {
                *reinterpret_cast<float2*>(pv + h) = *reinterpret_cast<const float2*>(p + i + h);
                *reinterpret_cast<float2*>(gv + h) = *reinterpret_cast<const float2*>(g + i + h);
                *reinterpret_cast<float2*>(mv + h) = *reinterpret_cast<const float2*>(m + i + h);
                *reinterpret_cast<float2*>(vv + h) = *reinterpret_cast<const float2*>(v + i + h);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { pv[e] = e < cnt ? p[i + e] : 0.f; gv[e] = e < cnt ? g[i + e] : 0.f; mv[e] = e < cnt ? m[i + e] : 0.f; vv[e] = e < cnt ? v[i + e] : 0.f; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            wisp_adamw_update(pv[e], mv[e], vv[e], gv[e] * gscale, lr, wd, b1, b2, eps, bc1, bc2_sqrt);
        }
        if (vec) {
            *reinterpret_cast<float4*>(p + i) = *reinterpret_cast<float4*>(pv);
            *reinterpret_cast<float4*>(m + i) = *reinterpret_cast<float4*>(mv);
            *reinterpret_cast<float4*>(v + i) = *reinterpret_cast<float4*>(vv);
            if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (vec2) {
#pragma unroll
            for (int h = 0; h < 4; h += 2) {
                *reinterpret_cast<float2*>(p + i + h) = *reinterpret_cast<float2*>(pv + h);
                *reinterpret_cast<float2*>(m + i + h) = *reinterpret_cast<float2*>(mv + h);
                *reinterpret_cast<float2*>(v + i + h) = *reinterpret_cast<float2*>(vv + h);
                if (zero_grad) *reinterpret_cast<float2*>(g + i + h) = make_float2(0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < cnt) { p[i + e] = pv[e]; m[i + e] = mv[e]; v[i + e] = vv[e]; if (zero_grad) g[i + e] = 0.0f; }
        }
        if (shadow) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < cnt) shadow[off + e] = __float2bfloat16(pv[e]);
        }
    }
}

extern "C" int wisp_adamw_step_groups(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int num_groups,
                                      const int64_t* group_begin, const int64_t* group_len, const float* group_lr,
                                      const float* group_weight_decay, void* const* group_bf16_shadow, float beta1,
                                      float beta2, float eps, int64_t step, float grad_scale, int zero_grad,
                                      wisp_stream_t stream) {
    WISP_REQUIRE(num_groups >= 1 && num_groups <= ADAMW_MAX_GROUPS && step >= 1, "bad group count / step");
    WISP_REQUIRE(param && grad && exp_avg && exp_avg_sq && group_begin && group_len && group_lr && group_weight_decay, "null pointer");
    WISP_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) &&
                 ((uintptr_t)exp_avg_sq % 16 == 0), "buffers must be 16-byte aligned");
    AdamGroups gr{};
    gr.n = num_groups;
    gr.chunk0[0] = 0;
    for (int k = 0; k < num_groups; ++k) {
        WISP_REQUIRE(group_begin[k] >= 0 && group_len[k] >= 0, "negative group range");
        gr.begin[k] = group_begin[k]; gr.len[k] = group_len[k];
        gr.lr[k] = group_lr[k]; gr.wd[k] = group_weight_decay[k];
        gr.shadow[k] = group_bf16_shadow ? (__hip_bfloat16*)group_bf16_shadow[k] : nullptr;
        gr.chunk0[k + 1] = gr.chunk0[k] + ceil_div64(group_len[k], 4);
    }
    if (gr.chunk0[num_groups] == 0) return WISP_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    const int grid = (int)min64(ceil_div64(gr.chunk0[num_groups], 256), 4096);
    hipLaunchKernelGGL(adamw_groups_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, const_cast<float*>(grad),
                       exp_avg, exp_avg_sq, gr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale, zero_grad);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---- the other optimizers BaseTrainer.init_optimizer can be configured with (wisp/config/presets/torch.py:45-68): Adam with
// coupled (L2) weight decay and RMSprop (nerf_octree.yaml:85, nerf_codebook.yaml:86 and the reference's best nerf_hash row,
// docs/pages/app_nerf.md:185-192).  Same flat-buffer / group / shadow / fused-zeroing structure as adamw_groups_kernel;
// arithmetic in torch.optim's order (torch/optim/adam.py _single_tensor_adam, rmsprop.py _single_tensor_rmsprop).
//   kind 0 AdamW  (decoupled decay)      s1 = exp_avg, s2 = exp_avg_sq            h = {beta1, beta2, -}
//   kind 1 Adam   (g += wd * p)          s1 = exp_avg, s2 = exp_avg_sq            h = {beta1, beta2, -}
//   kind 2 RMSprop (g += wd * p)         s1 = momentum buffer (momentum > 0), s2 = square_avg   h = {alpha, momentum, -}
template <int KIND>
__global__ void __launch_bounds__(256)
optim_groups_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1, float* __restrict__ s2,
                    AdamGroups gr, float h0, float h1, float eps, float bc1, float bc2_sqrt, float gscale, int zero_grad) {
    const int64_t total = gr.chunk0[gr.n];
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < ADAMW_MAX_GROUPS; ++q) k += (q < gr.n && c >= gr.chunk0[q]) ? 1 : 0;
        const int64_t off = (c - gr.chunk0[k]) * 4;
        const int64_t i = gr.begin[k] + off;
        const int cnt = (int)(gr.len[k] - off < 4 ? gr.len[k] - off : 4);
        const float lr = gr.lr[k], wd = gr.wd[k];
        __hip_bfloat16* shadow = gr.shadow[k];
        const bool use_s1 = (KIND != 2) || (h1 > 0.0f);
        float pv[4], gv[4], av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4];
        if (cnt == 4) {                                            // groups start on 16-byte boundaries
            *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(p + i);
            *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(g + i);
            *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(s2 + i);
            if (use_s1) *reinterpret_cast<float4*>(av) = *reinterpret_cast<const float4*>(s1 + i);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = e < cnt;
                pv[e] = in ? p[i + e] : 0.f; gv[e] = in ? g[i + e] : 0.f;
                av[e] = (in && use_s1) ? s1[i + e] : 0.f; bv[e] = in ? s2[i + e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = gv[e] * gscale;
            if (KIND == 0) { wisp_adamw_update(pv[e], av[e], bv[e], x, lr, wd, h0, h1, eps, bc1, bc2_sqrt); continue; }
            x += wd * pv[e];
            if (KIND == 2) {
                bv[e] = h0 * bv[e] + (1.0f - h0) * x * x;                  // square_avg
                const float avg = sqrtf(bv[e]) + eps;
                if (h1 > 0.0f) { av[e] = h1 * av[e] + x / avg; pv[e] -= lr * av[e]; }
                else pv[e] -= lr * (x / avg);
            } else {
                av[e] = h0 * av[e] + (1.0f - h0) * x;
                bv[e] = h1 * bv[e] + (1.0f - h1) * x * x;
                const float denom = sqrtf(bv[e]) / bc2_sqrt + eps;
                pv[e] -= (lr / bc1) * (av[e] / denom);
            }
        }
        if (cnt == 4) {
            *reinterpret_cast<float4*>(p + i) = *reinterpret_cast<float4*>(pv);
            *reinterpret_cast<float4*>(s2 + i) = *reinterpret_cast<float4*>(bv);
            if (use_s1) *reinterpret_cast<float4*>(s1 + i) = *reinterpret_cast<float4*>(av);
            if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < cnt) {
                    p[i + e] = pv[e]; s2[i + e] = bv[e];
                    if (use_s1) s1[i + e] = av[e];
                    if (zero_grad) g[i + e] = 0.0f;
                }
        }
        if (shadow) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < cnt) shadow[off + e] = __float2bfloat16(pv[e]);
        }
    }
}

extern "C" int wisp_optim_step_groups(int kind, float* param, const float* grad, float* state1, float* state2, int num_groups,
                                      const int64_t* group_begin, const int64_t* group_len, const float* group_lr,
                                      const float* group_weight_decay, void* const* group_bf16_shadow, float hyper0,
                                      float hyper1, float eps, int64_t step, float grad_scale, int zero_grad,
                                      wisp_stream_t stream) {
    WISP_REQUIRE(kind >= 0 && kind <= 2, "kind must be 0 (AdamW), 1 (Adam) or 2 (RMSprop)");
    WISP_REQUIRE(num_groups >= 1 && num_groups <= ADAMW_MAX_GROUPS && step >= 1, "bad group count / step");
    WISP_REQUIRE(param && grad && state2 && group_begin && group_len && group_lr && group_weight_decay, "null pointer");
    WISP_REQUIRE(state1 || (kind == 2 && hyper1 <= 0.0f), "state1 (exp_avg / momentum buffer) missing");
    WISP_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)state1 % 16 == 0) &&
                 ((uintptr_t)state2 % 16 == 0), "buffers must be 16-byte aligned");
    AdamGroups gr{};
    gr.n = num_groups;
    gr.chunk0[0] = 0;
    for (int k = 0; k < num_groups; ++k) {
        WISP_REQUIRE(group_begin[k] >= 0 && group_len[k] >= 0 && group_begin[k] % 4 == 0, "group ranges must start on a 16-byte boundary");
        gr.begin[k] = group_begin[k]; gr.len[k] = group_len[k];
        gr.lr[k] = group_lr[k]; gr.wd[k] = group_weight_decay[k];
        gr.shadow[k] = group_bf16_shadow ? (__hip_bfloat16*)group_bf16_shadow[k] : nullptr;
        gr.chunk0[k + 1] = gr.chunk0[k] + ceil_div64(group_len[k], 4);
    }
    if (gr.chunk0[num_groups] == 0) return WISP_OK;
    const float bc1 = kind == 2 ? 1.0f : 1.0f - powf(hyper0, (float)step);
    const float bc2_sqrt = kind == 2 ? 1.0f : sqrtf(1.0f - powf(hyper1, (float)step));
    const int grid = (int)min64(ceil_div64(gr.chunk0[num_groups], 256), 4096);
    float* gmut = const_cast<float*>(grad);
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0)
        hipLaunchKernelGGL(optim_groups_kernel<0>, dim3(grid), dim3(256), 0, s, param, gmut, state1, state2, gr, hyper0, hyper1, eps, bc1, bc2_sqrt, grad_scale, zero_grad);
    else if (kind == 1)
        hipLaunchKernelGGL(optim_groups_kernel<1>, dim3(grid), dim3(256), 0, s, param, gmut, state1, state2, gr, hyper0, hyper1, eps, bc1, bc2_sqrt, grad_scale, zero_grad);
    else
        hipLaunchKernelGGL(optim_groups_kernel<2>, dim3(grid), dim3(256), 0, s, param, gmut, state1, state2, gr, hyper0, hyper1, eps, bc1, bc2_sqrt, grad_scale, zero_grad);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---- ray-batch sampling: out_k[i, :] = src_k[index[i], :] for up to 4 row-major fp32 tensors that share ONE index vector
// (SampleRays, wisp/datasets/transforms/ray_sampler.py:25-35, picks the same random rays out of origins / dirs / rgb /
// further per-ray channels: one index_select launch each in the reference).  One launch, one pass over the index.
#define GATHER_MAX_SRCS 4
struct GatherSrcs { int n; const float* src[GATHER_MAX_SRCS]; float* dst[GATHER_MAX_SRCS]; int width[GATHER_MAX_SRCS]; };

__global__ void __launch_bounds__(256)
gather_rows_kernel(const int64_t* __restrict__ index, int64_t num, int64_t num_src_rows, GatherSrcs g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num) return;
    int64_t r = index[i];
    if (r < 0) r += num_src_rows;                                   // torch indexing semantics
    for (int k = 0; k < g.n; ++k) {
        const float* __restrict__ s = g.src[k] + r * g.width[k];
        float* __restrict__ d = g.dst[k] + i * g.width[k];
        for (int c = 0; c < g.width[k]; ++c) d[c] = s[c];
    }
}

extern "C" int wisp_gather_rows(const int64_t* index, int64_t num, int64_t num_src_rows, int num_tensors,
                                const float* const* src, const int* width, float* const* dst, wisp_stream_t stream) {
    WISP_REQUIRE(num >= 0 && num_src_rows >= 0 && num_tensors >= 1 && num_tensors <= GATHER_MAX_SRCS, "bad sizes");
    if (num == 0) return WISP_OK;
    WISP_REQUIRE(index && src && width && dst && num_src_rows > 0, "null pointer");
    GatherSrcs g{};
    g.n = num_tensors;
    for (int k = 0; k < num_tensors; ++k) {
        WISP_REQUIRE(src[k] && dst[k] && width[k] >= 1, "null tensor / bad width");
        g.src[k] = src[k]; g.dst[k] = dst[k]; g.width[k] = width[k];
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div64(num, 256)), dim3(256), 0, (hipStream_t)stream, index, num,
                       num_src_rows, g);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- small read-backs without a drain
// A variable-length op has to learn a count on the host before it can allocate its outputs (the reference syncs at the same
// places: nonzero in octree_as.py:288, the blocking cudaMemcpy of uniform_sample_cuda.cu:76).  Reading it on the compute stream
// would drain everything queued there, so the value travels on a side stream that only waits for the kernel that produced it:
// event on the compute stream -> side stream waits -> 8-byte copy to pinned memory -> event; the host later waits for that
// event alone.  This used to be five torch calls per step (two Event objects, a stream context, wait_event, copy_): 25 us of
// host time in a loop whose host side is critical at 2^18 samples per step.  One reader = one pinned word + two events + the
// side stream of its device; readers are pooled by the caller.
struct HostReader {
    int64_t* host;
    hipEvent_t ready, done;
    hipStream_t side;
    int device;
};
static hipStream_t g_reader_stream[64] = {nullptr};
static std::mutex g_reader_stream_lock;       // ctypes calls run without the GIL: two threads may create their first reader at once

// The reader binds the side stream of the device that is CURRENT here; the caller makes that the device of the tensors it will
// read (wisp/_C.py::_read_total_async switches devices around this call when they differ).
extern "C" void* wisp_host_reader_create(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    HostReader* r = new HostReader{nullptr, nullptr, nullptr, nullptr, dev};
    {
        std::lock_guard<std::mutex> hold(g_reader_stream_lock);
        if (!g_reader_stream[dev] && hipStreamCreateWithFlags(&g_reader_stream[dev], hipStreamNonBlocking) != hipSuccess) { delete r; return nullptr; }
        r->side = g_reader_stream[dev];
    }
    if (hipHostMalloc((void**)&r->host, sizeof(int64_t), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&r->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->done, hipEventDisableTiming) != hipSuccess) {
        if (r->host) (void)hipHostFree(r->host);
        if (r->ready) (void)hipEventDestroy(r->ready);
        delete r;
        return nullptr;
    }
    return r;
}

extern "C" int wisp_host_reader_issue(void* reader, const int64_t* src, wisp_stream_t stream) {
    WISP_REQUIRE(reader && src, "null pointer");
    HostReader* r = (HostReader*)reader;
    hipError_t e = hipEventRecord(r->ready, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(r->side, r->ready, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(r->host, src, sizeof(int64_t), hipMemcpyDeviceToHost, r->side);
    if (e == hipSuccess) e = hipEventRecord(r->done, r->side);
    if (e != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
    return WISP_OK;
}

extern "C" int wisp_host_reader_wait(void* reader, int64_t* value) {
    WISP_REQUIRE(reader && value, "null pointer");
    HostReader* r = (HostReader*)reader;
    const hipError_t e = hipEventSynchronize(r->done);
    if (e != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
    *value = *r->host;
    return WISP_OK;
}

extern "C" void wisp_host_reader_destroy(void* reader) {
    if (!reader) return;
    HostReader* r = (HostReader*)reader;
    (void)hipEventDestroy(r->ready);
    (void)hipEventDestroy(r->done);
    (void)hipHostFree(r->host);
    delete r;
}
