// Fused radiance-field decoder (placeholder translation unit: entry points exist so the ABI is complete;
// the MFMA implementation lands in the next milestone).
#include "wisp_common.h"

extern "C" int64_t wisp_nerf_mlp_param_count(int in_dim, int hidden, int view_freqs) {
    const int64_t pe = 3 + 6 * (int64_t)view_freqs;
    return (int64_t)hidden * in_dim + hidden + 16 * (int64_t)hidden + 16 + (int64_t)hidden * (15 + pe) + hidden +
           (int64_t)hidden * hidden + hidden + 3 * (int64_t)hidden + 3;
}

extern "C" int wisp_nerf_mlp_fwd(const void*, int, const float*, int64_t, int, int, int, const float*, int, float*, float*,
                                 wisp_stream_t) {
    return wisp_fail(WISP_ERR_UNSUPPORTED, __func__, "not built yet");
}

extern "C" int wisp_nerf_mlp_bwd(const void*, int, const float*, int64_t, int, int, int, const float*, int, const float*,
                                 const float*, void*, float*, wisp_stream_t) {
    return wisp_fail(WISP_ERR_UNSUPPORTED, __func__, "not built yet");
}
