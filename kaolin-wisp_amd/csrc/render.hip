// Packed (ragged) volume integration for gfx950.
//
// Replaces kaolin.render.spc.{cumsum, sum_reduce, exponential_integration} and the scatter block of
// PackedRFTracer.trace (wisp/tracers/packed_rf_tracer.py:143-165; semantics SURVEY.md Appendix A.4/A.5).
// The reference runs exp, a CUB scan, a multiply, three atomic segmented reductions and four index_put
// kernels; here one wave owns one pack (= one ray's samples): the exclusive transmittance sum is a 64-lane
// shuffle scan carried across 64-sample chunks, the per-ray colour / alpha / depth sums are wave
// reductions (deterministic, no atomics), and background blending + the scatter to per-ray buffers are
// fused into the same kernel.
#include "wisp_common.h"

// Wave-wide inclusive scan / sum on the VALU (DPP), no LDS traffic: Hillis-Steele inside each 16-lane row (row_shr 1, 2, 4, 8),
// then the row totals travel with row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3).  Lanes without a
// source add 0.  These kernels run one wave per ray and are bound by the LATENCY of their dependent chain - with
// ds_bpermute shuffles (~100+ cycles each, 30-50 per ray) that chain was most of the kernel.
#define WISP_DPP_ADD(V, CTRL, RMASK)                                                                                  \
    V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, RMASK, 0xf, false))
static __device__ __forceinline__ float wave_incl_scan_f(float v, int /*lane*/) {
    WISP_DPP_ADD(v, 0x111, 0xf);      // row_shr:1
    WISP_DPP_ADD(v, 0x112, 0xf);      // row_shr:2
    WISP_DPP_ADD(v, 0x114, 0xf);      // row_shr:4
    WISP_DPP_ADD(v, 0x118, 0xf);      // row_shr:8
    WISP_DPP_ADD(v, 0x142, 0xa);      // row_bcast:15 -> rows 1, 3
    WISP_DPP_ADD(v, 0x143, 0xc);      // row_bcast:31 -> rows 2, 3
    return v;
}
#undef WISP_DPP_ADD

static __device__ __forceinline__ float wave_last_f(float v) {      // value of lane 63, wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

static __device__ __forceinline__ float wave_sum_f(float v) { return wave_last_f(wave_incl_scan_f(v, 0)); }

// ---------------------------------------------------------------------------------------------- generic pack ops
__global__ void __launch_bounds__(256)
packed_sum_reduce_kernel(const float* __restrict__ feats, int64_t s_total, int channels,
                         const int64_t* __restrict__ pack_starts, int64_t num_packs, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= num_packs) return;
    const int64_t b = pack_starts[p], e = (p + 1 < num_packs) ? pack_starts[p + 1] : s_total;
    for (int c = 0; c < channels; ++c) {
        float acc = 0.0f;
        for (int64_t i = b + lane; i < e; i += 64) acc += feats[i * channels + c];
        acc = wave_sum_f(acc);
        if (lane == 0) out[p * channels + c] = acc;
    }
}

__global__ void __launch_bounds__(256)
packed_cumsum_kernel(const float* __restrict__ feats, int64_t s_total, int channels,
                     const int64_t* __restrict__ pack_starts, int64_t num_packs, int exclusive, int reverse,
                     float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= num_packs) return;
    const int64_t b = pack_starts[p], e = (p + 1 < num_packs) ? pack_starts[p + 1] : s_total;
    const int64_t len = e - b;
    for (int c = 0; c < channels; ++c) {
        float carry = 0.0f;
        for (int64_t k0 = 0; k0 < len; k0 += 64) {
            const int64_t k = k0 + lane;
            const int64_t i = reverse ? (e - 1 - k) : (b + k);
            const float v = (k < len) ? feats[i * channels + c] : 0.0f;
            const float inc = wave_incl_scan_f(v, lane);
            if (k < len) out[i * channels + c] = carry + (exclusive ? inc - v : inc);
            carry += wave_last_f(inc);
        }
    }
}

extern "C" int wisp_packed_sum_reduce(const float* feats, int64_t num_samples, int channels, const int64_t* pack_starts,
                                      int64_t num_packs, float* out, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0 && num_packs >= 0 && channels >= 1, "bad sizes");
    if (num_packs == 0) return WISP_OK;
    WISP_REQUIRE(feats && pack_starts && out, "null pointer");
    hipLaunchKernelGGL(packed_sum_reduce_kernel, dim3((unsigned)ceil_div64(num_packs, 4)), dim3(256), 0,
                       (hipStream_t)stream, feats, num_samples, channels, pack_starts, num_packs, out);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_packed_cumsum(const float* feats, int64_t num_samples, int channels, const int64_t* pack_starts,
                                  int64_t num_packs, int exclusive, int reverse, float* out, wisp_stream_t stream) {
    WISP_REQUIRE(num_samples >= 0 && num_packs >= 0 && channels >= 1, "bad sizes");
    if (num_packs == 0) return WISP_OK;
    WISP_REQUIRE(feats && pack_starts && out, "null pointer");
    hipLaunchKernelGGL(packed_cumsum_kernel, dim3((unsigned)ceil_div64(num_packs, 4)), dim3(256), 0,
                       (hipStream_t)stream, feats, num_samples, channels, pack_starts, num_packs, exclusive, reverse, out);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- fused compositing
struct Bg { float r, g, b; };

// One wave per ray, ONE wave per workgroup: these kernels are bound by the latency of a ray's dependent chain, and a
// workgroup of several rays holds its slots until its longest ray is done (4 -> 1 waves per workgroup: -2 us per kernel).
#define PACK_WAVES 1

__global__ void __launch_bounds__(256)
composite_init_kernel(int64_t num_rays, Bg bg, float* __restrict__ rgb, float* __restrict__ alpha,
                      float* __restrict__ depth, uint8_t* __restrict__ hit) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num_rays) return;
    rgb[r * 3] = bg.r; rgb[r * 3 + 1] = bg.g; rgb[r * 3 + 2] = bg.b;     // packed_rf_tracer.py:143
    alpha[r] = 0.0f;
    if (depth) depth[r] = 0.0f;
    hit[r] = 0;
}

// BY_RAY = false: pack p covers samples [pack_starts[p], pack_starts[p+1]) and its ray is ridx[first sample].
// BY_RAY = true : `pack_starts` is the per-ray offset table [R+1]; wave r handles ray r and also writes the
//                 background for rays without samples, so no init pass, no boundary->starts compaction and no
//                 device->host read of the pack count are needed.
template <bool BY_RAY>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(const float* __restrict__ color, const float* __restrict__ density,
                     const float* __restrict__ deltas, const float* __restrict__ depths,
                     const int64_t* __restrict__ ridx, const int64_t* __restrict__ pack_starts, int64_t num_packs,
                     int64_t s_total, Bg bg, float* __restrict__ out_rgb, float* __restrict__ out_alpha,
                     float* __restrict__ out_depth, uint8_t* __restrict__ out_hit, float* __restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= num_packs) return;
    const int64_t b = pack_starts[p], e = (BY_RAY || p + 1 < num_packs) ? pack_starts[p + 1] : s_total;
    if (BY_RAY && b == e) {
        if (lane == 0) {
            out_alpha[p] = 0.0f; out_hit[p] = 0;
            out_rgb[p * 3] = bg.r; out_rgb[p * 3 + 1] = bg.g; out_rgb[p * 3 + 2] = bg.b;
            if (out_depth) out_depth[p] = 0.0f;
        }
        return;
    }
    float carry = 0.0f, sr = 0.0f, sg = 0.0f, sb = 0.0f, sa = 0.0f, sd = 0.0f;
    for (int64_t k0 = b; k0 < e; k0 += 64) {
        const int64_t i = k0 + lane;
        const bool live = i < e;
        const float tau = live ? density[i] * deltas[i] : 0.0f;           // :152
        const float inc = wave_incl_scan_f(tau, lane);
        const float excl = carry + (inc - tau);
        carry += wave_last_f(inc);
        if (live) {
            const float T = expf(-excl);                                  // exponential_integration, exclusive=True
            const float w = T * (1.0f - expf(-tau));
            weights[i] = w;
            sr += w * color[i * 3]; sg += w * color[i * 3 + 1]; sb += w * color[i * 3 + 2];
            sa += w;
            if (depths) sd += w * depths[i];
        }
    }
    sr = wave_sum_f(sr); sg = wave_sum_f(sg); sb = wave_sum_f(sb); sa = wave_sum_f(sa);
    if (depths) sd = wave_sum_f(sd);
    if (lane == 0) {
        const int64_t r = BY_RAY ? p : ridx[b];
        out_alpha[r] = sa;                                                // :160-161
        out_hit[r] = sa > 0.0f ? 1 : 0;                                   // :162
        const float om = 1.0f - sa;
        out_rgb[r * 3] = bg.r * om + sr; out_rgb[r * 3 + 1] = bg.g * om + sg; out_rgb[r * 3 + 2] = bg.b * om + sb;  // :165
        if (out_depth) out_depth[r] = sd;                                 // :157-158
    }
}

// dL/dc_i = w_i g_rgb ; G_i = g_rgb.(c_i - bg) + g_alpha + g_depth t_i ;
// dL/dtau_i = G_i T_i exp(-tau_i) - sum_{k>i} G_k w_k ; dL/dsigma_i = dL/dtau_i * delta_i
template <bool BY_RAY>
__global__ void __launch_bounds__(256)
composite_bwd_kernel(const float* __restrict__ grad_rgb, const float* __restrict__ grad_alpha,
                     const float* __restrict__ grad_depth, const float* __restrict__ color,
                     const float* __restrict__ density, const float* __restrict__ deltas,
                     const float* __restrict__ depths, const int64_t* __restrict__ ridx,
                     const int64_t* __restrict__ pack_starts, int64_t num_packs, int64_t s_total, Bg bg,
                     float* __restrict__ grad_color, float* __restrict__ grad_density) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= num_packs) return;
    const int64_t b = pack_starts[p], e = (BY_RAY || p + 1 < num_packs) ? pack_starts[p + 1] : s_total;
    if (BY_RAY && b == e) return;
    const int64_t r = BY_RAY ? p : ridx[b];
    const float gr = grad_rgb[r * 3], gg = grad_rgb[r * 3 + 1], gb = grad_rgb[r * 3 + 2];
    const float ga = grad_alpha ? grad_alpha[r] : 0.0f;
    const float gd = (grad_depth && depths) ? grad_depth[r] : 0.0f;
    // pass 1: total of G_k w_k over the pack
    float carry = 0.0f, tot = 0.0f;
    for (int64_t k0 = b; k0 < e; k0 += 64) {
        const int64_t i = k0 + lane;
        const bool live = i < e;
        const float tau = live ? density[i] * deltas[i] : 0.0f;
        const float inc = wave_incl_scan_f(tau, lane);
        const float excl = carry + (inc - tau);
        carry += wave_last_f(inc);
        if (live) {
            const float w = expf(-excl) * (1.0f - expf(-tau));
            float G = gr * (color[i * 3] - bg.r) + gg * (color[i * 3 + 1] - bg.g) + gb * (color[i * 3 + 2] - bg.b) + ga;
            if (depths) G += gd * depths[i];
            tot += G * w;
        }
    }
    tot = wave_sum_f(tot);
    // pass 2: gradients, suffix = total - inclusive prefix of G w
    carry = 0.0f;
    float gcarry = 0.0f;
    for (int64_t k0 = b; k0 < e; k0 += 64) {
        const int64_t i = k0 + lane;
        const bool live = i < e;
        const float dl = live ? deltas[i] : 0.0f;
        const float tau = live ? density[i] * dl : 0.0f;
        const float inc = wave_incl_scan_f(tau, lane);
        const float excl = carry + (inc - tau);
        carry += wave_last_f(inc);
        float T = 0.0f, et = 0.0f, w = 0.0f, G = 0.0f;
        if (live) {
            T = expf(-excl); et = expf(-tau); w = T * (1.0f - et);
            G = gr * (color[i * 3] - bg.r) + gg * (color[i * 3 + 1] - bg.g) + gb * (color[i * 3 + 2] - bg.b) + ga;
            if (depths) G += gd * depths[i];
        }
        const float gw = G * w;
        const float ginc = wave_incl_scan_f(gw, lane);
        const float suffix = tot - (gcarry + ginc);
        gcarry += wave_last_f(ginc);
        if (live) {
            grad_color[i * 3] = w * gr; grad_color[i * 3 + 1] = w * gg; grad_color[i * 3 + 2] = w * gb;
            grad_density[i] = (G * T * et - suffix) * dl;
        }
    }
}

extern "C" int wisp_composite_fwd(const float* color, const float* density, const float* deltas, const float* depths,
                                  const int64_t* ridx, const int64_t* pack_starts, int64_t num_packs,
                                  int64_t num_samples, int64_t num_rays, const float* bg, float* out_rgb,
                                  float* out_alpha, float* out_depth, uint8_t* out_hit, float* weights,
                                  wisp_stream_t stream) {
    WISP_REQUIRE(num_rays >= 0 && num_packs >= 0 && num_samples >= 0, "bad sizes");
    WISP_REQUIRE(bg, "null bg");
    if (num_rays == 0) return WISP_OK;
    WISP_REQUIRE(out_rgb && out_alpha && out_hit, "null output");
    WISP_REQUIRE(out_depth == nullptr || depths != nullptr, "out_depth needs depths");
    hipStream_t s = (hipStream_t)stream;
    const Bg b{bg[0], bg[1], bg[2]};
    if (ridx == nullptr) {      // per-ray offsets mode: pack_starts = ray_offsets [num_rays + 1], num_packs must equal num_rays
        WISP_REQUIRE(pack_starts && num_packs == num_rays, "ray-offset mode needs offsets[R+1] and num_packs == num_rays");
        WISP_REQUIRE(num_samples == 0 || (color && density && deltas && weights), "null pointer");
        hipLaunchKernelGGL(composite_fwd_kernel<true>, dim3((unsigned)ceil_div64(num_rays, PACK_WAVES)), dim3(64 * PACK_WAVES), 0, s, color,
                           density, deltas, out_depth ? depths : nullptr, ridx, pack_starts, num_rays, num_samples, b,
                           out_rgb, out_alpha, out_depth, out_hit, weights);
        WISP_CHECK_LAUNCH();
        return WISP_OK;
    }
    hipLaunchKernelGGL(composite_init_kernel, dim3((unsigned)ceil_div64(num_rays, 256)), dim3(256), 0, s, num_rays, b,
                       out_rgb, out_alpha, out_depth, out_hit);
    if (num_packs > 0) {
        WISP_REQUIRE(color && density && deltas && ridx && pack_starts && weights, "null pointer");
        hipLaunchKernelGGL(composite_fwd_kernel<false>, dim3((unsigned)ceil_div64(num_packs, 4)), dim3(256), 0, s, color,
                           density, deltas, out_depth ? depths : nullptr, ridx, pack_starts, num_packs, num_samples, b,
                           out_rgb, out_alpha, out_depth, out_hit, weights);
    }
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

extern "C" int wisp_composite_bwd(const float* grad_rgb, const float* grad_alpha, const float* grad_depth,
                                  const float* color, const float* density, const float* deltas, const float* depths,
                                  const int64_t* ridx, const int64_t* pack_starts, int64_t num_packs,
                                  int64_t num_samples, const float* bg, float* grad_color, float* grad_density,
                                  wisp_stream_t stream) {
    WISP_REQUIRE(num_packs >= 0 && num_samples >= 0 && bg, "bad sizes");
    if (num_packs == 0) return WISP_OK;
    WISP_REQUIRE(grad_rgb && color && density && deltas && pack_starts && grad_color && grad_density, "null pointer");
    const Bg b{bg[0], bg[1], bg[2]};
    if (ridx == nullptr)        // per-ray offsets mode (see wisp_composite_fwd): num_packs = number of rays
        hipLaunchKernelGGL(composite_bwd_kernel<true>, dim3((unsigned)ceil_div64(num_packs, PACK_WAVES)), dim3(64 * PACK_WAVES), 0,
                           (hipStream_t)stream, grad_rgb, grad_alpha, grad_depth, color, density, deltas, depths, ridx,
                           pack_starts, num_packs, num_samples, b, grad_color, grad_density);
    else
        hipLaunchKernelGGL(composite_bwd_kernel<false>, dim3((unsigned)ceil_div64(num_packs, 4)), dim3(256), 0,
                           (hipStream_t)stream, grad_rgb, grad_alpha, grad_depth, color, density, deltas, depths, ridx,
                           pack_starts, num_packs, num_samples, b, grad_color, grad_density);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- sphere-tracing helper
// find_depth_bound (wisp/csrc/render/find_depth_bound_cuda.cu:16-45): for every ray (pack) advance its current nugget
// index to the first nugget whose [entry, exit] contains, or lies beyond, the query depth; -1 when there is none.
// Bug-compatible with the reference on purpose: the search of pack i is bounded by the CURRENT index of pack i+1
// (not that pack's first nugget), and the last pack is bounded by num_packs rather than num_nugs (.cu:29).
// One deliberate difference: a finished neighbour (-1) makes that bound 0xFFFFFFFF and the reference then reads past the end
// of `depth` when the query lies beyond every remaining nugget (undefined there); here the scan also stops at num_nugs -> -1.
__global__ void __launch_bounds__(256)
find_depth_bound_kernel(int64_t num_packs, int64_t num_nugs, const float* __restrict__ query,
                        const int32_t* __restrict__ curr_in, int32_t* __restrict__ curr_out,
                        const float* __restrict__ depth) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= num_packs) return;
    int32_t result = -1;
    const int32_t start = curr_in[t];
    if (start > -1) {
        uint32_t i = (uint32_t)start;
        uint32_t stop = (t == num_packs - 1) ? (uint32_t)num_packs : (uint32_t)curr_in[t + 1];
        if ((int64_t)stop > num_nugs) stop = (uint32_t)num_nugs;
        const float q = query[t];
        while (i < stop) {
            const float entry = depth[2 * (int64_t)i], exit_ = depth[2 * (int64_t)i + 1];
            if ((q >= entry && q <= exit_) || q < entry) { result = (int32_t)i; break; }
            ++i;
        }
    }
    curr_out[t] = result;
}

extern "C" int wisp_find_depth_bound(const float* query, const int32_t* curr_idxes, const float* nug_depth,
                                     int64_t num_packs, int64_t num_nugs, int32_t* out, wisp_stream_t stream) {
    WISP_REQUIRE(num_packs >= 0 && num_nugs >= 0, "bad sizes");
    if (num_packs == 0) return WISP_OK;
    WISP_REQUIRE(query && curr_idxes && nug_depth && out, "null pointer");
    hipLaunchKernelGGL(find_depth_bound_kernel, dim3((unsigned)ceil_div64(num_packs, 256)), dim3(256), 0, (hipStream_t)stream,
                       num_packs, num_nugs, query, curr_idxes, out, nug_depth);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- fused sphere-trace step
// One iteration of PackedSDFTracer.trace's marching loop (wisp/tracers/packed_sdf_tracer.py:118-146) for every active
// ray in ONE launch: advance t by the last distance, convergence tests, far-plane test, nugget search
// (find_depth_bound, same bounds quirks as above), jump to the next cell's entry, and the new query position.
// The reference issues ~25 masked torch kernels per iteration for this.
// State per pack p (= ray with at least one nugget), all updated in place except curr_idx which is double-buffered
// because pack p reads pack p+1's PREVIOUS index as its search bound:
//   t, dist, dist_prev f32 ; mask, hit u8 ; curr_in/curr_out i32 ; curr_pidx i64 ; x f32[3]
__global__ void __launch_bounds__(256)
sphere_trace_step_kernel(int64_t num_packs, const float* __restrict__ nug_o, const float* __restrict__ nug_d,
                         const float* __restrict__ nug_depth, const int32_t* __restrict__ nug_pidx, float dist_max,
                         float thr_close, float thr_avg, float* __restrict__ t, const float* __restrict__ dist, float* __restrict__ dist_prev,
                         uint8_t* __restrict__ mask, uint8_t* __restrict__ hit, const int32_t* __restrict__ curr_in,
                         int32_t* __restrict__ curr_out, int64_t* __restrict__ curr_pidx, float* __restrict__ x) {
#pragma clang fp contract(off)
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= num_packs) return;
    const int32_t cur = curr_in[p];
    bool m = mask[p] != 0;
    bool h = hit[p] != 0;
    const float dd = dist[p];
    float tt = t[p] + dd;                                                        // t += dist          (:120)
    if (m) {
        h = fabsf(dd) < thr_close;                                               // :122   thr_close = min_dis
        h = h || (fabsf(dd + dist_prev[p]) * 0.5f < thr_avg);                    // :123-124 thr_avg = 5 min_dis
        m = tt < dist_max;                                                       // :125
    }
    m = m && !h;                                                                 // :126
    if (m) dist_prev[p] = dd;                                                    // :129
    // find_depth_bound (:131) - evaluated for every pack with a valid index, exactly like the reference kernel
    int32_t nxt = -1;
    if (cur > -1) {
        uint32_t i = (uint32_t)cur;
        const uint32_t stop = (p == num_packs - 1) ? (uint32_t)num_packs : (uint32_t)curr_in[p + 1];
        while (i < stop) {
            const float entry = nug_depth[2 * (int64_t)i], exit_ = nug_depth[2 * (int64_t)i + 1];
            if ((tt >= entry && tt <= exit_) || tt < entry) { nxt = (int32_t)i; break; }
            ++i;
        }
    }
    m = m && (nxt != -1);                                                        // :132
    const bool jumped = nxt != cur;                                              // :133
    const int32_t now = m ? nxt : cur;                                           // :134
    if (m && jumped) tt = nug_depth[2 * (int64_t)now];                           // :136
    if (m) {                                                                     // :137-139
        x[p * 3 + 0] = nug_o[p * 3 + 0] + nug_d[p * 3 + 0] * tt;
        x[p * 3 + 1] = nug_o[p * 3 + 1] + nug_d[p * 3 + 1] * tt;
        x[p * 3 + 2] = nug_o[p * 3 + 2] + nug_d[p * 3 + 2] * tt;
        curr_pidx[p] = (int64_t)nug_pidx[now];
    } else if (mask[p] != 0) {
        // ray deactivated in this step: x still has to be refreshed with the advanced t (:121 ran before the tests)
        x[p * 3 + 0] = nug_o[p * 3 + 0] + nug_d[p * 3 + 0] * tt;
        x[p * 3 + 1] = nug_o[p * 3 + 1] + nug_d[p * 3 + 1] * tt;
        x[p * 3 + 2] = nug_o[p * 3 + 2] + nug_d[p * 3 + 2] * tt;
    }
    t[p] = tt;
    mask[p] = m ? 1 : 0;
    hit[p] = h ? 1 : 0;
    curr_out[p] = now;
}

extern "C" int wisp_sphere_trace_step(int64_t num_packs, const float* nug_o, const float* nug_d, const float* nug_depth,
                                      const int32_t* nug_pidx, float dist_max, float thr_close, float thr_avg, float* t,
                                      const float* dist,
                                      float* dist_prev, uint8_t* mask, uint8_t* hit, const int32_t* curr_in,
                                      int32_t* curr_out, int64_t* curr_pidx, float* x, wisp_stream_t stream) {
    WISP_REQUIRE(num_packs >= 0, "negative count");
    if (num_packs == 0) return WISP_OK;
    WISP_REQUIRE(nug_o && nug_d && nug_depth && nug_pidx && t && dist && dist_prev && mask && hit && curr_in && curr_out &&
                 curr_pidx && x, "null pointer");
    hipLaunchKernelGGL(sphere_trace_step_kernel, dim3((unsigned)ceil_div64(num_packs, 256)), dim3(256), 0, (hipStream_t)stream,
                       num_packs, nug_o, nug_d, nug_depth, nug_pidx, dist_max, thr_close, thr_avg, t, dist, dist_prev, mask, hit, curr_in,
                       curr_out, curr_pidx, x);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- ray generation
// generate_pinhole_rays / generate_ortho_rays (wisp/ops/raygen/raygen.py:40-119) for one camera: pixel coordinates ->
// principal-point shift -> NDC -> camera-space ray -> world space (inverse of the view transform: R^T (p - t)) ->
// normalised direction.  Every step is a separately rounded fp32 operation, in the reference's order.
struct RayCam {
    float x0, y0, width, height;      // principal point offset (pixels from the image centre), image size
    float sx, sy;                     // pinhole: tan(fov_x / 2), tan(fov_y / 2); ortho: fov_distance * aspect, fov_distance
    float r[9];                       // view rotation R (row major), world -> camera
    float t[3];                       // view translation
};

template <bool ORTHO>
__global__ void __launch_bounds__(256)
raygen_kernel(const float* __restrict__ pixel_x, const float* __restrict__ pixel_y, int64_t n, RayCam cam,
              float* __restrict__ origins, float* __restrict__ dirs) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float px = pixel_x[i], py = pixel_y[i];
    if (!ORTHO) { px = px - cam.x0; py = py + cam.y0; }                  // raygen.py:66-67
    px = 2.0f * (px / cam.width) - 1.0f;                                  // _to_ndc_coords, :34-37
    py = 2.0f * (py / cam.height) - 1.0f;
    float o[3], d[3];
    if (ORTHO) {                                                          // :100-107
        o[0] = px * cam.sx; o[1] = -(py * cam.sy); o[2] = 0.0f;
        d[0] = 0.0f; d[1] = 0.0f; d[2] = -1.0f;
    } else {                                                              // :72-77
        o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
        d[0] = px * cam.sx; d[1] = -py * cam.sy; d[2] = -1.0f;
    }
    // inv_transform_rays: origin' = R^T (o - t), dir' = R^T d  (sums accumulated left to right)
    const float q[3] = {o[0] - cam.t[0], o[1] - cam.t[1], o[2] - cam.t[2]};
    float ow[3], dw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ow[c] = (cam.r[0 + c] * q[0] + cam.r[3 + c] * q[1]) + cam.r[6 + c] * q[2];
        dw[c] = (cam.r[0 + c] * d[0] + cam.r[3 + c] * d[1]) + cam.r[6 + c] * d[2];
    }
    const float nrm = sqrtf((dw[0] * dw[0] + dw[1] * dw[1]) + dw[2] * dw[2]);        // torch.linalg.norm
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        origins[i * 3 + c] = ow[c];
        dirs[i * 3 + c] = dw[c] / nrm;
    }
}

extern "C" int wisp_generate_rays(const float* pixel_x, const float* pixel_y, int64_t num_pixels, int ortho, float x0,
                                  float y0, float width, float height, float scale_x, float scale_y,
                                  const float* view_rotation, const float* view_translation, float* origins, float* dirs,
                                  wisp_stream_t stream) {
    WISP_REQUIRE(num_pixels >= 0, "negative count");
    if (num_pixels == 0) return WISP_OK;
    WISP_REQUIRE(pixel_x && pixel_y && view_rotation && view_translation && origins && dirs, "null pointer");
    WISP_REQUIRE(width > 0.0f && height > 0.0f, "bad image size");
    RayCam cam;
    cam.x0 = x0; cam.y0 = y0; cam.width = width; cam.height = height; cam.sx = scale_x; cam.sy = scale_y;
    for (int k = 0; k < 9; ++k) cam.r[k] = view_rotation[k];             // host arrays
    for (int k = 0; k < 3; ++k) cam.t[k] = view_translation[k];
    const dim3 grid((unsigned)ceil_div64(num_pixels, 256));
    if (ortho)
        hipLaunchKernelGGL(raygen_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, pixel_x, pixel_y, num_pixels, cam, origins, dirs);
    else
        hipLaunchKernelGGL(raygen_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, pixel_x, pixel_y, num_pixels, cam, origins, dirs);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- photometric loss
// mean over all elements of huber (smooth-L1, beta = 1) / L2 / L1 between the composited colours and the ground truth
// (MultiviewTrainer.step, wisp/trainers/multiview_trainer.py:140-154) together with its gradient w.r.t. the colours:
// d loss / d rgb = clamp(x, -1, 1) / N, 2 x / N, sign(x) / N for x = rgb - gt.  A grid-stride pass writes the gradient and one
// partial sum per workgroup; a one-workgroup second launch adds the partials in index order - reproducible value.
// (Until round 3 the workgroup that finished last did that inside the first launch - device-scope fence + ticket per workgroup.
//  On this part such a fence writes back and invalidates an XCD's L2; see composite_loss_kernel for what 8192 of them cost.)
#define LOSS_BLOCKS 256
__global__ void __launch_bounds__(256)
rgb_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ gt, int64_t n, int kind, float inv_n,
                float* __restrict__ grad, float* __restrict__ partial) {
    __shared__ float part[4];
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = rgb[i] - gt[i];
        float l, g;
        if (kind == 0) { const float a = fabsf(x); l = a < 1.0f ? 0.5f * x * x : a - 0.5f; g = fminf(fmaxf(x, -1.0f), 1.0f); }
        else if (kind == 1) { l = x * x; g = 2.0f * x; }
        else { l = fabsf(x); g = (x > 0.0f) ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
        acc += l;
        grad[i] = g * inv_n;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ void __launch_bounds__(1024)
loss_sum_kernel(const float* __restrict__ partial, int n, float inv_n, float* __restrict__ loss) {
    __shared__ float part[16];
    float t = 0.0f;
    for (int base = 0; base < n; base += 8 * 1024) {      // eight loads of a thread in flight at once
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = base + threadIdx.x + k * 1024; v[k] = i < n ? partial[i] : 0.0f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += part[w];
        loss[0] = s * inv_n;
    }
}

extern "C" int wisp_rgb_loss(const float* rgb, const float* gt, int64_t num_elements, int kind, float* grad, float* loss,
                             float* workspace, wisp_stream_t stream) {
    WISP_REQUIRE(num_elements > 0 && rgb && gt && grad && loss && workspace, "bad arguments");
    WISP_REQUIRE(kind >= 0 && kind <= 2, "kind: 0 huber, 1 l2, 2 l1");
    const int blocks = (int)min64(ceil_div64(num_elements, 256), LOSS_BLOCKS);
    const float inv_n = 1.0f / (float)num_elements;
    hipLaunchKernelGGL(rgb_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rgb, gt, num_elements, kind, inv_n, grad,
                       workspace + 1);
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, workspace + 1, blocks, inv_n, loss);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}

// ---------------------------------------------------------------------------------------------- composite + loss + backward
// The three launches a training step makes between the decoder's forward and backward - compositing (packed_rf_tracer.py:143-165),
// the photometric loss with its gradient (multiview_trainer.py:140-154) and the compositing backward - as ONE pass per ray: a
// wave composites its ray, forms the loss terms and d loss / d rgb of that ray on the spot and runs the backward recurrences
// while the ray's samples are still in registers (rays of at most 64 samples, the usual case) or in cache (longer rays: the
// same two extra passes as composite_bwd_kernel).  What the separate kernels wrote only for each other never reaches memory:
// the weights, rgb / alpha / hit and the colour gradient per ray.  Same arithmetic, same order, per ray; the loss is the sum of
// per-workgroup partials added in index order by a one-wave second launch (reproducible), its terms are grouped per ray
// instead of per thread-strided element, so its value agrees with wisp_rgb_loss to rounding, not bit for bit.
// W waves per workgroup, each wave a "group" of its own (group v walks rays v, v + groups, ...: which rays a partial sum covers,
// and hence the loss bits, do not depend on W).  W = 4: a quarter of the workgroups to dispatch for the same waves.
template <int W>
__global__ void __launch_bounds__(64 * W)
composite_loss_kernel(const float* __restrict__ color, const float* __restrict__ density, const float* __restrict__ deltas,
                      const int64_t* __restrict__ offsets, int64_t num_rays, int groups, Bg bg, const float* __restrict__ gt, int kind,
                      float inv_n, float* __restrict__ grad_color, float* __restrict__ grad_density, float* __restrict__ out_rgb,
                      float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * W + (threadIdx.x >> 6);
    if (v >= groups) return;
    float lacc = 0.0f;
    for (int64_t r = v; r < num_rays; r += groups) {
        const int64_t b = offsets[r], e = offsets[r + 1];
        // the ray's ground truth does not depend on anything computed below: ask for it now, not after the reductions
        const float gt0 = gt[r * 3], gt1 = gt[r * 3 + 1], gt2 = gt[r * 3 + 2];
        const bool single = e - b <= 64;
        float carry = 0.0f, sr = 0.0f, sg = 0.0f, sb = 0.0f, sa = 0.0f;
        float dl = 0.0f, T = 0.0f, et = 0.0f, w = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;     // the single chunk's values
        for (int64_t k0 = b; k0 < e; k0 += 64) {
            const int64_t i = k0 + lane;
            const bool live = i < e;
            // every load of the chunk is issued before the scan: with the colours fetched behind it (where they are first used) a
            // ray paid one more memory round trip in its dependent chain
            dl = live ? deltas[i] : 0.0f;
            const float dn = live ? density[i] : 0.0f;
            c0 = live ? color[i * 3] : 0.0f; c1 = live ? color[i * 3 + 1] : 0.0f; c2 = live ? color[i * 3 + 2] : 0.0f;
            const float tau = live ? dn * dl : 0.0f;
            const float inc = wave_incl_scan_f(tau, lane);
            const float excl = carry + (inc - tau);
            carry += wave_last_f(inc);
            T = 0.0f; et = 0.0f; w = 0.0f;
            if (live) {
                T = expf(-excl); et = expf(-tau);
                w = T * (1.0f - et);
                sr += w * c0; sg += w * c1; sb += w * c2;
                sa += w;
            }
        }
        if (b < e) { sr = wave_sum_f(sr); sg = wave_sum_f(sg); sb = wave_sum_f(sb); sa = wave_sum_f(sa); }
        const float om = 1.0f - sa;
        const float rgb[3] = {bg.r * om + sr, bg.g * om + sg, bg.b * om + sb};
        const float gtc[3] = {gt0, gt1, gt2};
        float g[3], lsum[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = rgb[c] - gtc[c];
            float l, d;
            if (kind == 0) { const float a = fabsf(x); l = a < 1.0f ? 0.5f * x * x : a - 0.5f; d = fminf(fmaxf(x, -1.0f), 1.0f); }
            else if (kind == 1) { l = x * x; d = 2.0f * x; }
            else { l = fabsf(x); d = (x > 0.0f) ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
            lsum[c] = l;
            g[c] = d * inv_n;
        }
        lacc += (lsum[0] + lsum[1]) + lsum[2];
        if (out_rgb && lane == 0) { out_rgb[r * 3] = rgb[0]; out_rgb[r * 3 + 1] = rgb[1]; out_rgb[r * 3 + 2] = rgb[2]; }
        if (b == e) continue;
        const float gr = g[0], gg = g[1], gb = g[2];
        if (single) {
            const int64_t i = b + lane;
            const bool live = i < e;
            const float G = live ? gr * (c0 - bg.r) + gg * (c1 - bg.g) + gb * (c2 - bg.b) : 0.0f;
            const float gw = G * w;
            const float ginc = wave_incl_scan_f(gw, lane);
            const float tot = wave_last_f(ginc);
            const float suffix = tot - ginc;
            if (live) {
                grad_color[i * 3] = w * gr; grad_color[i * 3 + 1] = w * gg; grad_color[i * 3 + 2] = w * gb;
                grad_density[i] = (G * T * et - suffix) * dl;
            }
            continue;
        }
        // longer rays: as composite_bwd_kernel - the total of G w first, then the gradients with the suffix sums
        carry = 0.0f;
        float tot = 0.0f;
        for (int64_t k0 = b; k0 < e; k0 += 64) {
            const int64_t i = k0 + lane;
            const bool live = i < e;
            const float tau = live ? density[i] * deltas[i] : 0.0f;
            const float inc = wave_incl_scan_f(tau, lane);
            const float excl = carry + (inc - tau);
            carry += wave_last_f(inc);
            if (live) {
                const float ww = expf(-excl) * (1.0f - expf(-tau));
                const float G = gr * (color[i * 3] - bg.r) + gg * (color[i * 3 + 1] - bg.g) + gb * (color[i * 3 + 2] - bg.b);
                tot += G * ww;
            }
        }
        tot = wave_sum_f(tot);
        carry = 0.0f;
        float gcarry = 0.0f;
        for (int64_t k0 = b; k0 < e; k0 += 64) {
            const int64_t i = k0 + lane;
            const bool live = i < e;
            const float dd = live ? deltas[i] : 0.0f;
            const float tau = live ? density[i] * dd : 0.0f;
            const float inc = wave_incl_scan_f(tau, lane);
            const float excl = carry + (inc - tau);
            carry += wave_last_f(inc);
            float TT = 0.0f, ee = 0.0f, ww = 0.0f, G = 0.0f;
            if (live) {
                TT = expf(-excl); ee = expf(-tau); ww = TT * (1.0f - ee);
                G = gr * (color[i * 3] - bg.r) + gg * (color[i * 3 + 1] - bg.g) + gb * (color[i * 3 + 2] - bg.b);
            }
            const float gw = G * ww;
            const float ginc = wave_incl_scan_f(gw, lane);
            const float suffix = tot - (gcarry + ginc);
            gcarry += wave_last_f(ginc);
            if (live) {
                grad_color[i * 3] = ww * gr; grad_color[i * 3 + 1] = ww * gg; grad_color[i * 3 + 2] = ww * gb;
                grad_density[i] = (G * TT * ee - suffix) * dd;
            }
        }
    }
    // loss: one partial per workgroup (every lane holds the same sum); loss_sum_kernel adds them in index order.  (A last-
    // workgroup-finishes reduction inside this kernel - device-scope fence + ticket per workgroup - took the launch from 36 to
    // 238 us at 38 K rays: on this part a device-scope release / acquire writes back and invalidates the XCD's L2, 8192 times.)
    if (lane == 0) partial[v] = lacc;
}

extern "C" int wisp_composite_loss(const float* color, const float* density, const float* deltas, const int64_t* ray_offsets,
                                   int64_t num_rays, int64_t num_samples, const float* bg, const float* gt, int kind,
                                   float* grad_color, float* grad_density, float* out_rgb, float* loss, float* workspace,
                                   int64_t workspace_floats, wisp_stream_t stream) {
    WISP_REQUIRE(num_rays > 0 && num_samples >= 0 && bg && gt && ray_offsets && loss && workspace && workspace_floats >= 1, "bad arguments");
    WISP_REQUIRE(kind >= 0 && kind <= 2, "kind: 0 huber, 1 l2, 2 l1");
    WISP_REQUIRE(num_samples == 0 || (color && density && deltas && grad_color && grad_density), "null pointer");
    const Bg b{bg[0], bg[1], bg[2]};
    // one ray per wave when the workspace has a partial-sum cell for every ray (rays differ a lot in length: handing a wave
    // several of them in a fixed order leaves the launch waiting for the unluckiest wave), else a grid-stride walk
    const int groups = (int)min64(num_rays, min64(workspace_floats, (int64_t)1 << 22));
    const float inv_n = 1.0f / (float)(num_rays * 3);
    static const int waves = [] { const char* e = getenv("WISP_COMPOSITE_WAVES"); const int w = e ? atoi(e) : 4; return (w == 1 || w == 2) ? w : 4; }();
    if (waves == 1)
        hipLaunchKernelGGL(composite_loss_kernel<1>, dim3(groups), dim3(64), 0, (hipStream_t)stream, color, density, deltas, ray_offsets,
                           num_rays, groups, b, gt, kind, inv_n, grad_color, grad_density, out_rgb, workspace);
    else if (waves == 2)
        hipLaunchKernelGGL(composite_loss_kernel<2>, dim3((groups + 1) / 2), dim3(128), 0, (hipStream_t)stream, color, density, deltas, ray_offsets,
                           num_rays, groups, b, gt, kind, inv_n, grad_color, grad_density, out_rgb, workspace);
    else
        hipLaunchKernelGGL(composite_loss_kernel<4>, dim3((groups + 3) / 4), dim3(256), 0, (hipStream_t)stream, color, density, deltas, ray_offsets,
                           num_rays, groups, b, gt, kind, inv_n, grad_color, grad_density, out_rgb, workspace);
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, workspace, groups, inv_n, loss);
    WISP_CHECK_LAUNCH();
    return WISP_OK;
}
