// One optimisation step of the nerf_hash.yaml training shape issued from native code: every launch of
//   MultiviewTrainer.step (wisp/trainers/multiview_trainer.py:111-180) -> PackedRFTracer.trace (packed_rf_tracer.py:84-181) ->
//   HashGrid.interpolate + NeuralRadianceField.rgba (nerf.py:219-264) -> compositing + loss -> backward -> AdamW (base_trainer.py:205-246)
// - ray emit, the next batch's occupancy count + scan + size read-back, hash-grid lookup, view code, decoder forward, compositing +
// loss + its backward, decoder backward, hash-grid backward (with the table's AdamW folded in), AdamW of everything else - in ONE
// call across the language boundary.  No kernel lives here: the function calls the library's own C-ABI entry points in the order
// wisp/trainers/multiview_trainer.py::_DirectNeRFStep.run issues them, on buffers carved out of one caller-owned workspace, so the
// results are the Python-issued step's, bit for bit (tests/test_gpu_1_selfcheck.py).
//
// What it buys (measured, profiles/r06_native_step_ab.txt): the interpreter needs ~0.2 ms to issue a step through ctypes (13 calls, 19
// allocations, ~70 pointer extractions; scripts/prof_step_host.py).  At the reference trainer's 2^18 packed samples per step the kernels
// take ~0.30 ms and the step 0.34-0.38 ms: the host waits for the look-ahead read-back most of the time, i.e. it is NOT the bottleneck
// there - the gaps of the kernel trace are the device's own launch-to-launch latency (16 dependent launches) - and the native step
// is neutral (0.377 vs 0.377 ms in alternating bench runs, 1.10 vs 1.10 ms at 2^21).  Below ~2^17 samples per step the interpreter
// becomes the longer of the two and the native step wins: 0.266 -> 0.251 ms at 2^16, 0.254 -> 0.232 ms at 2^15 (what one GPU of a
// strong-scaled 8-GPU run at the reference's global batch would see).  It also takes the interpreter's jitter (collector pauses, a
// busy host) out of the step.  A HIP graph would need every kernel to read the sample count from device memory; issuing from C++ needs
// nothing of the kind - the count of a batch is read back one step ahead and is a host value by the time its step starts.
#include "wisp_common.h"

#include <new>
#include <vector>

namespace {

constexpr int NS_TIMED_KERNELS = 4;                   // hashgrid_fwd, nerf_mlp_fwd, nerf_mlp_bwd, hashgrid_bwd
constexpr int NS_MAX_TIMED_STEPS = 512;                // event pairs are created with the handle: recording must not pay for them
constexpr int NS_LOSS_RING = 8;

struct MarchSlot {
    const float* origins = nullptr;
    const float* dirs = nullptr;
    int64_t num_rays = 0;
    uint64_t seed = 0;
    uint32_t* hitmask = nullptr;
    int32_t* counts = nullptr;
    int64_t* offsets = nullptr;
    void* scan_ws = nullptr;
    void* reader = nullptr;
    bool counted = false;
};

struct Layout {
    int64_t hitmask[2], counts[2], offsets[2], scan_ws[2];
    int64_t ridx, samples, depth, deltas, boundary, feats, dir_code, color, density, g_color, g_density, g_feats, comp_ws, loss, mlp_ws;
    int64_t comp_ws_floats, mlp_ws_bytes, total;
};

static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

static int elt_bytes(int dtype) { return dtype == WISP_F32 ? 4 : 2; }

static Layout layout_of(const wisp_nerf_step_config& c) {
    Layout L{};
    int64_t at = 0;
    auto take = [&](int64_t bytes) { const int64_t o = at; at = align256(at + bytes); return o; };
    const int64_t R = c.max_rays, S = c.max_samples;
    const int64_t words = (c.num_samples + 31) / 32;
    for (int k = 0; k < 2; ++k) {
        L.hitmask[k] = take(R * words * 4);
        L.counts[k] = take(R * 4);
        L.offsets[k] = take((R + 1) * 8);
        L.scan_ws[k] = take(wisp_scan_workspace_bytes(R));
    }
    const int64_t row = (int64_t)c.num_lods * c.feature_dim * elt_bytes(c.dtype_table);
    L.ridx = take(S * 8);
    L.samples = take(S * 12);
    L.depth = take(S * 4);
    L.deltas = take(S * 4);
    L.boundary = take(S);
    L.feats = take(S * row);
    L.dir_code = take(R * 32 * 2);
    L.color = take(S * 12);
    L.density = take(S * 4);
    L.g_color = take(S * 12);
    L.g_density = take(S * 4);
    L.g_feats = take(S * row);
    L.comp_ws_floats = R > 4096 ? 2 * R : 8192;
    L.comp_ws = take(L.comp_ws_floats * 4);
    L.loss = take(NS_LOSS_RING * 4);
    L.mlp_ws_bytes = wisp_nerf_mlp_bwd_workspace_bytes(S, c.hidden) + 256;
    L.mlp_ws = take(L.mlp_ws_bytes);
    L.total = at;
    return L;
}

struct Step {
    wisp_nerf_step_config cfg;
    std::vector<int64_t> first_idx_host;
    std::vector<int32_t> res;
    char* ws = nullptr;
    Layout lay{};
    MarchSlot slot[2];
    int loss_at = 0;
    // live timing of the four roofline entry points (bench.py): event pairs, created on first use
    std::vector<hipEvent_t> ev;
    std::vector<int64_t> ev_units;
    int timed_steps = 0;
};

template <typename T> static T* at(const Step* s, int64_t off) { return reinterpret_cast<T*>(s->ws + off); }

static bool config_ok(const wisp_nerf_step_config* c, const char** why) {
#define NEED(cond, text) do { if (!(cond)) { *why = text; return false; } } while (0)
    NEED(c != nullptr, "null config");
    NEED(c->struct_bytes == (int64_t)sizeof(wisp_nerf_step_config), "wisp_nerf_step_config has another size than this library's (header / binding drift)");
    NEED(c->octree && c->exsum && c->level >= 1 && c->level <= 10 && c->num_samples >= 1, "occupancy structure");
    NEED(c->range > 0.0f, "dist_max must exceed dist_min");
    NEED(c->table_lookup && c->first_idx && c->first_idx_host && c->resolutions, "hash table");
    NEED(c->num_lods >= 1 && c->num_lods <= 16 && c->feature_dim == 2, "the fused step covers two-feature tables of at most 16 levels");
    NEED(c->dtype_table == WISP_BF16 || c->dtype_table == WISP_F16, "the fused step reads a 16-bit copy of the table (bf16 / fp16)");
    NEED(c->in_dim == c->num_lods * c->feature_dim && c->in_dim <= 32 && c->hidden == 64 && c->view_freqs == 4, "decoder shape (hidden 64, 4 view frequencies, 'cat' features)");
    NEED(c->dec_params && c->dec_grad && c->table_grad, "parameter / gradient pointers");
    NEED(c->loss_kind >= 0 && c->loss_kind <= 2, "loss kind");
    NEED(c->max_rays >= 1 && c->max_samples >= 1, "capacities");
    NEED(c->flat_param && c->flat_grad && c->flat_exp_avg && c->flat_exp_avg_sq, "flat optimizer buffers");
#undef NEED
    return true;
}

static int do_count(Step* s, int k, const float* o, const float* d, int64_t R, uint64_t seed, hipStream_t st) {
    const wisp_nerf_step_config& c = s->cfg;
    MarchSlot& m = s->slot[k];
    m.counted = false;
    if (!o || !d || R < 1) return wisp_fail(WISP_ERR_INVALID, "wisp_nerf_step_count", "null rays");
    if (R > c.max_rays) return wisp_fail(WISP_ERR_CAPACITY, "wisp_nerf_step_count", "more rays than the step was created for");
    int rc = wisp_raymarch_ray_count(c.occ_bits, c.octree, c.exsum, o, d, R, c.near, c.range, c.num_samples, c.level, nullptr, seed,
                                     c.coarse_bits, c.coarse_level, m.hitmask, m.counts, st);
    if (rc != WISP_OK) return rc;
    rc = wisp_exclusive_scan_i32(m.counts, R, m.offsets, m.scan_ws, st);
    if (rc != WISP_OK) return rc;
    rc = wisp_host_reader_issue(m.reader, m.offsets + R, st);
    if (rc != WISP_OK) return rc;
    m.origins = o; m.dirs = d; m.num_rays = R; m.seed = seed; m.counted = true;
    return WISP_OK;
}

}  // namespace

extern "C" int64_t wisp_nerf_step_config_bytes(void) { return (int64_t)sizeof(wisp_nerf_step_config); }

extern "C" int64_t wisp_nerf_step_workspace_bytes(const wisp_nerf_step_config* cfg) {
    const char* why = "";
    if (!config_ok(cfg, &why)) return wisp_fail(WISP_ERR_INVALID, __func__, why);
    return layout_of(*cfg).total + 256;
}

extern "C" void* wisp_nerf_step_create(const wisp_nerf_step_config* cfg, void* workspace, int64_t workspace_bytes) {
    const char* why = "";
    if (!config_ok(cfg, &why)) { (void)wisp_fail(WISP_ERR_INVALID, __func__, why); return nullptr; }
    Step* s = new (std::nothrow) Step();
    if (!s) return nullptr;
    s->cfg = *cfg;
    s->first_idx_host.assign(cfg->first_idx_host, cfg->first_idx_host + cfg->num_lods + 1);
    s->res.assign(cfg->resolutions, cfg->resolutions + cfg->num_lods);
    s->cfg.first_idx_host = s->first_idx_host.data();
    s->cfg.resolutions = s->res.data();
    s->lay = layout_of(s->cfg);
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
    if (!workspace || base + s->lay.total > reinterpret_cast<char*>(workspace) + workspace_bytes) {
        (void)wisp_fail(WISP_ERR_INVALID, __func__, "workspace smaller than wisp_nerf_step_workspace_bytes()");
        delete s;
        return nullptr;
    }
    s->ws = base;
    for (int k = 0; k < 2; ++k) {
        MarchSlot& m = s->slot[k];
        m.hitmask = at<uint32_t>(s, s->lay.hitmask[k]);
        m.counts = at<int32_t>(s, s->lay.counts[k]);
        m.offsets = at<int64_t>(s, s->lay.offsets[k]);
        m.scan_ws = at<void>(s, s->lay.scan_ws[k]);
        m.reader = wisp_host_reader_create();
        if (!m.reader) {
            (void)wisp_fail(WISP_ERR_LAUNCH, __func__, "wisp_host_reader_create failed");
            if (k == 1) wisp_host_reader_destroy(s->slot[0].reader);
            delete s;
            return nullptr;
        }
    }
    // the timing events exist from the start (hipEventCreate inside a timed step would be part of what it measures)
    s->ev.reserve((size_t)NS_MAX_TIMED_STEPS * 2 * NS_TIMED_KERNELS);
    for (int i = 0; i < NS_MAX_TIMED_STEPS * 2 * NS_TIMED_KERNELS; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        s->ev.push_back(e);
    }
    s->ev_units.assign(NS_MAX_TIMED_STEPS, 0);
    return s;
}

// The same handle over other buffers of the same shapes - what a prune does to a trainer (a new octree, new bitfields): workspace,
// read-back slots and timing events stay; batches counted against the old octree are dropped (count them again).
extern "C" int wisp_nerf_step_reconfigure(void* step, const wisp_nerf_step_config* cfg) {
    WISP_REQUIRE(step, "null handle");
    const char* why = "";
    if (!config_ok(cfg, &why)) return wisp_fail(WISP_ERR_INVALID, __func__, why);
    Step* s = static_cast<Step*>(step);
    const Layout now = layout_of(*cfg);
    if (now.total != s->lay.total || cfg->max_rays != s->cfg.max_rays || cfg->max_samples != s->cfg.max_samples ||
        cfg->num_samples != s->cfg.num_samples || cfg->num_lods != s->cfg.num_lods || cfg->dtype_table != s->cfg.dtype_table ||
        cfg->hidden != s->cfg.hidden)
        return wisp_fail(WISP_ERR_INVALID, __func__, "another workspace layout: destroy the handle and create a new one");
    for (int k = 0; k < 2; ++k)
        if (s->slot[k].counted) { int64_t v = 0; (void)wisp_host_reader_wait(s->slot[k].reader, &v); s->slot[k].counted = false; }
    s->cfg = *cfg;
    s->first_idx_host.assign(cfg->first_idx_host, cfg->first_idx_host + cfg->num_lods + 1);
    s->res.assign(cfg->resolutions, cfg->resolutions + cfg->num_lods);
    s->cfg.first_idx_host = s->first_idx_host.data();
    s->cfg.resolutions = s->res.data();
    return WISP_OK;
}

extern "C" void wisp_nerf_step_destroy(void* step) {
    if (!step) return;
    Step* s = static_cast<Step*>(step);
    for (int k = 0; k < 2; ++k) {
        if (s->slot[k].counted) { int64_t v = 0; (void)wisp_host_reader_wait(s->slot[k].reader, &v); }     // the copy targets the reader's word
        wisp_host_reader_destroy(s->slot[k].reader);
    }
    for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
    delete s;
}

extern "C" int wisp_nerf_step_count(void* step, int slot, const float* origins, const float* dirs, int64_t num_rays, uint64_t seed,
                                    wisp_stream_t stream) {
    WISP_REQUIRE(step && (slot == 0 || slot == 1), "bad handle / slot");
    Step* s = static_cast<Step*>(step);
    if (s->slot[slot].counted) { int64_t v = 0; (void)wisp_host_reader_wait(s->slot[slot].reader, &v); }      // an abandoned count: let its copy land
    return do_count(s, slot, origins, dirs, num_rays, seed, (hipStream_t)stream);
}

extern "C" int wisp_nerf_step_run(void* step, int slot, const float* gts, const float* next_origins, const float* next_dirs,
                                  int64_t next_num_rays, uint64_t next_seed, const wisp_nerf_step_hyper* hp,
                                  const float* level_cap_scale, void* hashgrid_workspace, int64_t hashgrid_workspace_bytes,
                                  int record_timing, int64_t* num_samples, int64_t* covered_rows, float** loss, wisp_stream_t stream) {
    WISP_REQUIRE(step && (slot == 0 || slot == 1) && gts && hp && num_samples && loss, "null argument");
    Step* s = static_cast<Step*>(step);
    const wisp_nerf_step_config& c = s->cfg;
    MarchSlot& m = s->slot[slot];
    WISP_REQUIRE(m.counted, "this slot holds no counted batch (wisp_nerf_step_count first)");
    WISP_REQUIRE(hp->struct_bytes == (int64_t)sizeof(wisp_nerf_step_hyper), "wisp_nerf_step_hyper has another size than this library's");
    hipStream_t st = (hipStream_t)stream;
    int64_t S = 0;
    int rc = wisp_host_reader_wait(m.reader, &S);
    m.counted = false;
    if (rc != WISP_OK) return rc;
    *num_samples = S;
    const int64_t R = m.num_rays;
    // (a batch the buffers cannot hold, or an empty one: the caller's modular step takes it - the count can be redone from the seed)
    if (S > c.max_samples) return wisp_fail(WISP_ERR_CAPACITY, __func__, "more packed samples than the step was created for");
    if (S < 1) return wisp_fail(WISP_ERR_CAPACITY, __func__, "no sample survived the occupancy test");

    int64_t* ridx = at<int64_t>(s, s->lay.ridx);
    float* samples = at<float>(s, s->lay.samples);
    float* depth = at<float>(s, s->lay.depth);
    float* deltas = at<float>(s, s->lay.deltas);
    uint8_t* boundary = at<uint8_t>(s, s->lay.boundary);
    void* feats = at<void>(s, s->lay.feats);
    void* code = at<void>(s, s->lay.dir_code);
    float* color = at<float>(s, s->lay.color);
    float* density = at<float>(s, s->lay.density);
    float* g_color = at<float>(s, s->lay.g_color);
    float* g_density = at<float>(s, s->lay.g_density);
    void* g_feats = at<void>(s, s->lay.g_feats);
    float* loss_slot = at<float>(s, s->lay.loss) + s->loss_at;
    s->loss_at = (s->loss_at + 1) % NS_LOSS_RING;
    *loss = loss_slot;

    hipEvent_t* ev = nullptr;
    if (record_timing && s->timed_steps < NS_MAX_TIMED_STEPS &&
        s->ev.size() >= (size_t)(s->timed_steps + 1) * 2 * NS_TIMED_KERNELS) {          // (a full record: later steps go untimed)
        ev = s->ev.data() + (size_t)s->timed_steps * 2 * NS_TIMED_KERNELS;
        s->ev_units[s->timed_steps] = S;
        ++s->timed_steps;
    }
#define NS_MARK(i) do { if (ev) (void)hipEventRecord(ev[i], st); } while (0)
#define NS_CALL(expr) do { rc = (expr); if (rc != WISP_OK) return rc; } while (0)

    // ---- this batch's samples; the NEXT batch's occupancy test right behind them (its size read-back is long done when the next
    //      step asks for it)
    NS_CALL(wisp_raymarch_ray_emit(m.origins, m.dirs, R, c.near, c.range, c.num_samples, nullptr, m.seed, m.hitmask, m.offsets, ridx,
                                   samples, depth, deltas, boundary, nullptr, st));
    if (next_origins) NS_CALL(do_count(s, slot ^ 1, next_origins, next_dirs, next_num_rays, next_seed, st));

    // ---- forward
    NS_MARK(0);
    NS_CALL(wisp_hashgrid_interpolate_fwd(samples, S, 3, c.table_lookup, c.dtype_table, c.feature_dim, c.first_idx, c.resolutions,
                                          c.num_lods, c.bitwidth, c.zero_from_col, feats, st));
    NS_MARK(1);
    NS_CALL(wisp_nerf_mlp_dir_code(m.dirs, R, c.view_freqs, code, st));
    NS_MARK(2);
    NS_CALL(wisp_nerf_mlp_fwd_rays(feats, c.dtype_table, code, ridx, S, c.in_dim, c.hidden, c.view_freqs, c.dec_params, color, density, st));
    NS_MARK(3);
    NS_CALL(wisp_composite_loss(color, density, deltas, m.offsets, R, S, c.bg, gts, c.loss_kind, g_color, g_density, nullptr, loss_slot,
                                at<float>(s, s->lay.comp_ws), s->lay.comp_ws_floats, st));
    // ---- backward
    NS_MARK(4);
    NS_CALL(wisp_nerf_mlp_bwd_rays(feats, c.dtype_table, code, ridx, S, c.in_dim, c.hidden, c.view_freqs, c.dec_params, g_color, g_density,
                                   g_feats, c.dec_grad, at<float>(s, s->lay.mlp_ws), s->lay.mlp_ws_bytes, st));
    NS_MARK(5);
    int64_t covered[16] = {0};
    const bool fold = hp->optimizer == 2 && c.table_param && c.table_exp_avg && c.table_exp_avg_sq;
    NS_MARK(6);
    if (fold)
        NS_CALL(wisp_hashgrid_interpolate_bwd_adamw(samples, S, 3, g_feats, c.dtype_table, c.feature_dim, c.first_idx, c.resolutions,
                                                    c.num_lods, c.bitwidth, c.zero_from_col, c.table_grad, hashgrid_workspace,
                                                    hashgrid_workspace_bytes, level_cap_scale, c.table_param, c.table_exp_avg,
                                                    c.table_exp_avg_sq, c.table_shadow, hp->lr_grid, hp->beta1, hp->beta2, hp->eps,
                                                    hp->weight_decay, hp->step, hp->grad_scale, covered, st));
    else
        NS_CALL(wisp_hashgrid_interpolate_bwd(samples, S, 3, g_feats, c.dtype_table, c.feature_dim, c.first_idx, c.resolutions, c.num_lods,
                                              c.bitwidth, c.zero_from_col, c.table_grad, hashgrid_workspace, hashgrid_workspace_bytes,
                                              level_cap_scale, st));
    NS_MARK(7);
    if (covered_rows)
        for (int l = 0; l < c.num_lods; ++l) covered_rows[l] = covered[l];

    // ---- AdamW of everything the folded flush did not take: decoder, 'rest', and the grid rows no reduce workgroup owned
    //      (MultiviewTrainStep.optimizer_step / _uncovered_grid_ranges: the same ranges in the same order, one launch)
    if (hp->optimizer >= 1) {
        int64_t begin[24], len[24];
        float lr[24], wd[24];
        void* shadow[24];
        int n = 0;
        auto add = [&](int64_t b, int64_t l, float rate, void* sh) {
            if (l > 0 && n < 24) { begin[n] = b; len[n] = l; lr[n] = rate; wd[n] = hp->weight_decay; shadow[n] = sh; ++n; }
        };
        add(c.decoder_begin, c.decoder_len, hp->lr_decoder, nullptr);
        const int64_t ga = c.grid_begin, gb = c.grid_begin + c.grid_len;
        char* gsh = static_cast<char*>(c.grid_shadow);
        if (!fold) {
            add(ga, c.grid_len, hp->lr_grid, gsh);
        } else {
            int64_t at_ = ga;
            for (int l = 0; l < c.num_lods; ++l) {
                int64_t rows = covered[l];
                const int64_t have = s->first_idx_host[l + 1] - s->first_idx_host[l];
                if (rows > have) rows = have;
                if (rows <= 0) continue;
                const int64_t lo = c.table_offset + s->first_idx_host[l] * c.feature_dim;
                const int64_t hi = c.table_offset + (s->first_idx_host[l] + rows) * c.feature_dim;
                if (lo > at_) add(at_, lo - at_, hp->lr_grid, gsh ? gsh + (at_ - ga) * 2 : nullptr);
                if (hi > at_) at_ = hi;
            }
            if (gb > at_) add(at_, gb - at_, hp->lr_grid, gsh ? gsh + (at_ - ga) * 2 : nullptr);
        }
        add(c.rest_begin, c.rest_len, hp->lr_rest, nullptr);
        if (n > 0)
            NS_CALL(wisp_adamw_step_groups(c.flat_param, c.flat_grad, c.flat_exp_avg, c.flat_exp_avg_sq, n, begin, len, lr, wd, shadow,
                                           hp->beta1, hp->beta2, hp->eps, hp->step, hp->grad_scale, 1, st));
    }
#undef NS_MARK
#undef NS_CALL
    return WISP_OK;
}

extern "C" int wisp_nerf_step_read_timing(void* step, int max_steps, float* ms /* host [max_steps][4] */, int64_t* units /* host [max_steps] */,
                                          int* num_steps /* host */) {
    WISP_REQUIRE(step && ms && units && num_steps && max_steps >= 0, "null argument");
    Step* s = static_cast<Step*>(step);
    const int n = s->timed_steps < max_steps ? s->timed_steps : max_steps;
    for (int i = 0; i < n; ++i) {
        hipEvent_t* ev = s->ev.data() + (size_t)i * 2 * NS_TIMED_KERNELS;
        for (int k = 0; k < NS_TIMED_KERNELS; ++k) {
            float t = 0.0f;
            const hipError_t e = hipEventElapsedTime(&t, ev[2 * k], ev[2 * k + 1]);      // (needs the events complete: synchronise first)
            if (e != hipSuccess) return wisp_fail(WISP_ERR_LAUNCH, __func__, hipGetErrorString(e));
            ms[(size_t)i * NS_TIMED_KERNELS + k] = t;
        }
        units[i] = s->ev_units[i];
    }
    *num_steps = n;
    s->timed_steps = 0;
    return WISP_OK;
}
