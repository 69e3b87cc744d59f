"""Host side of wisp_nerf_step_* (csrc/train_step.hip): the nerf_hash.yaml training step issued by ONE call into the library.

MultiviewTrainStep.step hands a batch here when the pipeline has the shape the native step covers (see `applies`); everything the
Python-issued direct step (`_DirectNeRFStep.run`) does per iteration - emit, look-ahead count, lookup, decoder, compositing + loss,
backward, the table's AdamW in the flush, AdamW of the rest - then happens inside `wisp_nerf_step_run`, on the same kernels with the
same arguments (same results bit for bit: tests/test_gpu_1_selfcheck.py::test_native_step_*).  What stays here is bookkeeping the
library cannot know: which octree is current (a prune replaces it: the handle is rebuilt and the pending count redone from its
seed), the learned record-slot scales of the hash-grid backward, the learning-rate schedule, and the fall-back to the Python-issued
step for a batch the preallocated buffers cannot hold.  WISP_NATIVE_STEP=0 switches it off."""
import ctypes
import os

import numpy as np
import torch

ENABLED = os.environ.get("WISP_NATIVE_STEP", "1") != "0"


def _hip():
    import wisp._C as C
    return C


class _Elapsed:
    """bench.py reads its timing sink as (start, end, units) with start.elapsed_time(end): a measured duration in that shape"""
    __slots__ = ("ms",)

    def __init__(self, ms):
        self.ms = ms

    def elapsed_time(self, _other):
        return self.ms


class NativeHashStep:
    TIMED = ("hashgrid_fwd", "nerf_mlp_fwd", "nerf_mlp_bwd", "hashgrid_bwd")

    def __init__(self, trainer):
        self.t = trainer
        self.handle = None
        self.key = None
        self.shape = None
        self.keep = None              # tensors / ctypes arrays the handle borrows
        self.pending = None           # dict(rays, slot, seed, blas) of the batch counted ahead
        self.next_slot = 0
        self.hg_ws = None
        self.hg_scale = None
        self.fallbacks = 0
        self.steps = 0

    # ------------------------------------------------------------------------------------------------ applicability
    def applies(self, rays, jitter):
        t = self.t
        d = t._direct
        C = _hip()
        if not ENABLED or d is None or not d.hash_fast or not t.enable_amp or d.biasless or jitter is not None:
            return False
        if t.optimizer != 'adamw' or t.world > 1 or t.force_allreduce or getattr(t, "grad_accum_steps", 1) > 1:
            return False
        if t.comm_timing is not None or C.TIMING_ALL is not None:
            return False
        cls = type(t)
        from wisp.trainers.multiview_trainer import MultiviewTrainStep, FUSED_COMPOSITE_LOSS
        if not FUSED_COMPOSITE_LOSS or 'optimizer_step' in t.__dict__ or cls.optimizer_step is not MultiviewTrainStep.optimizer_step \
                or cls.reduce_and_update is not MultiviewTrainStep.reduce_and_update:
            return False
        pipe = t.pipeline
        tracer, grid = pipe.tracer, pipe.nef.grid
        i, h, f = d.shape
        if tracer.raymarch_type != 'ray' or h != 64 or f != 4 or i != len(d.res) * 2 or i > 32 or d.table.shape[1] != 2:
            return False
        if t.rgb_loss_type not in ('huber', 'l2', 'l1') or not rays.origins.is_cuda or rays.origins.dtype != torch.float32:
            return False
        if torch.is_tensor(rays.dist_min) or torch.is_tensor(rays.dist_max):
            return False
        if not C.nerf_mlp_rays_preferred(torch.bfloat16, i, h, f, True):
            return False
        blas = grid.blas
        return blas.max_level <= 10 and rays.origins.shape[0] <= t.max_rays and rays.origins.is_contiguous() and rays.dirs.is_contiguous()

    # ------------------------------------------------------------------------------------------------ handle
    def _capacity(self):
        t = self.t
        return int(t.max_rays), int(max(2 * t.target_sample_size, 1 << 17))

    def _ensure(self, rays):
        """(re)build the handle when anything it borrows has changed: the octree (prune), the ray interval, the buffers, the capacity"""
        C = _hip()
        t, d = self.t, self.t._direct
        from wisp.ops.grid import current_shadow
        pipe = t.pipeline
        tracer, grid = pipe.tracer, pipe.nef.grid
        blas = grid.blas
        dev = rays.origins.device
        blas._to_device(dev)
        f = t.flat
        shadow = current_shadow(d.table, torch.bfloat16)
        if shadow is None:
            return False                              # (a torch-side write outdated the bf16 copy: the Python step casts the master)
        packed, packed_grad = d._params()
        max_rays, max_samples = self._capacity()
        near, far = float(rays.dist_min), float(rays.dist_max)
        key = (id(blas), blas.octree.data_ptr(), near, far, tracer.num_steps, d.table.data_ptr(), shadow.data_ptr(), f.data.data_ptr(),
               f.grad.data_ptr(), packed.data_ptr(), packed_grad.data_ptr(), max_rays, max_samples, t.rgb_loss_type,
               tuple(float(v) for v in tracer._bg_host()), torch.cuda.current_stream().cuda_stream)
        if self.handle is not None and key == self.key:
            return True
        level = blas.max_level
        occ = blas._bitfield(level)
        if occ is None:
            return False
        coarse, lc = blas._coarse_bitfield(rays, tracer.num_steps, level)
        L = len(d.res)
        first_host = (ctypes.c_int64 * (L + 1))(*[int(v) for v in d._first_idx_host[:L + 1]])
        res = (ctypes.c_int32 * L)(*d.res)
        off = next(o for p, o in f._grid_params if p is d.table)
        n = d.table.numel()
        ga, gb = f.ranges["grid"]
        da, db = f.ranges["decoder"]
        ra, rb = f.ranges["rest"]
        cfg = C.NerfStepConfig()
        cfg.struct_bytes = ctypes.sizeof(C.NerfStepConfig)
        P = lambda x: None if x is None else x.data_ptr()
        cfg.occ_bits, cfg.octree, cfg.exsum, cfg.coarse_bits = P(occ), P(blas.octree), P(blas.prefix), P(coarse)
        cfg.table_lookup, cfg.first_idx = P(shadow), P(d.first_idx)
        cfg.first_idx_host, cfg.resolutions = ctypes.cast(first_host, ctypes.c_void_p), ctypes.cast(res, ctypes.c_void_p)
        cfg.table_param, cfg.table_grad = P(d.table), P(d.table.grad)
        cfg.table_exp_avg, cfg.table_exp_avg_sq = f.exp_avg.data_ptr() + 4 * off, f.exp_avg_sq.data_ptr() + 4 * off
        cfg.table_shadow = None if f.shadow is None else f.shadow.data_ptr() + 2 * (off - ga)
        cfg.dec_params, cfg.dec_grad = P(packed), P(packed_grad)
        cfg.flat_param, cfg.flat_grad, cfg.flat_exp_avg, cfg.flat_exp_avg_sq = P(f.data), P(f.grad), P(f.exp_avg), P(f.exp_avg_sq)
        cfg.grid_shadow = P(f.shadow)
        cfg.decoder_begin, cfg.decoder_len, cfg.grid_begin, cfg.grid_len, cfg.rest_begin, cfg.rest_len = da, db - da, ga, gb - ga, ra, rb - ra
        cfg.table_offset, cfg.max_rays, cfg.max_samples = off, max_rays, max_samples
        cfg.level, cfg.coarse_level, cfg.num_samples = level, int(lc), int(tracer.num_steps)
        cfg.loss_kind = {"huber": 0, "l2": 1, "l1": 2}[t.rgb_loss_type]
        cfg.dtype_table, cfg.num_lods, cfg.feature_dim, cfg.bitwidth = C.BF16, L, 2, int(d.bitwidth)
        i, h, fq = d.shape
        cfg.zero_from_col, cfg.in_dim, cfg.hidden, cfg.view_freqs = int(d.zero_from_col), i, h, fq
        cfg.near = float(np.float32(near))
        cfg.range = float(np.float32(far - near))                     # depth *= (dist_max - dist_min), python double -> f32
        bg = tracer._bg_host()
        cfg.bg[0], cfg.bg[1], cfg.bg[2] = float(bg[0]), float(bg[1]), float(bg[2])
        shape = (max_rays, max_samples, int(tracer.num_steps), L, dev, torch.cuda.current_stream().cuda_stream)
        if self.handle is not None and shape == self.shape and C.lib.wisp_nerf_step_reconfigure(self.handle, ctypes.byref(cfg)) == 0:
            ws = self.keep[0]                 # same shapes over other buffers (a prune): workspace, read-back slots, events stay
        else:
            self.close()
            need = int(C.lib.wisp_nerf_step_workspace_bytes(ctypes.byref(cfg)))
            if need <= 0:
                raise RuntimeError(f"wisp_nerf_step_workspace_bytes: {C.last_error()}")
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self.handle = C.lib.wisp_nerf_step_create(ctypes.byref(cfg), ws.data_ptr(), need)
            if not self.handle:
                self.handle = None
                raise RuntimeError(f"wisp_nerf_step_create: {C.last_error()}")
            self.loss_view = {}
        self.key, self.shape = key, shape
        self.keep = (ws, first_host, res, occ, coarse, shadow, packed, packed_grad, blas, cfg)
        self.blas = blas
        # a batch counted into the OLD handle's buffers is gone: it is redone from its seed (same jitter stream) when its step comes
        if self.pending is not None:
            self.pending["slot"] = None
        return True

    def close(self):
        if self.handle is not None:
            _hip().lib.wisp_nerf_step_destroy(self.handle)
            self.handle, self.key, self.keep, self.shape = None, None, None, None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------ one step
    def _hyper(self):
        C = _hip()
        t = self.t
        step = t.opt_steps + 1
        s = t._lr_scale(step)
        hp = C.NerfStepHyper()
        hp.struct_bytes = ctypes.sizeof(C.NerfStepHyper)
        hp.step = step
        hp.lr_decoder = t.lr * s
        hp.lr_grid = t.lr * t.grid_lr_weight * s
        hp.lr_rest = t.lr * s
        hp.weight_decay, hp.beta1, hp.beta2, hp.eps = t.weight_decay, t.betas[0], t.betas[1], t.eps
        hp.grad_scale = 1.0 / t.world
        hp.optimizer = 2 if t.fuse_grid_optimizer else 1
        return hp

    def _hashgrid_scratch(self, dev, n_guess):
        """record-slot scales + scratch of the hash-grid backward for launches of up to max_samples samples"""
        C = _hip()
        d = self.t._direct
        res_key, res_arr, res_ptr = C._host_res(d.res)
        fit = C._slot_fit(dev, 3, C.BF16, 2, res_key, d.bitwidth, d.zero_from_col, n_guess >= C.HASHGRID_EMIT_WIDE_MIN) if n_guess >= 4096 else None
        scale_arr, scale_ptr = fit.scales(n_guess) if fit is not None else (None, None)
        if self.hg_ws is None or self.hg_scale is not scale_arr:
            _, max_samples = self._capacity()
            need = max(int(C.lib.wisp_hashgrid_bwd_workspace_bytes(n, 3, C.BF16, 2, res_ptr, len(d.res), d.bitwidth, scale_ptr))
                       for n in (max_samples, min(max_samples, C.HASHGRID_EMIT_WIDE_MIN - 1), max(4096, int(1.3 * n_guess))))
            if self.hg_ws is None or self.hg_ws.numel() < need or self.hg_ws.numel() > 3 * need + (64 << 20):
                self.hg_ws = None
                self.hg_ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
            self.hg_scale = scale_arr
        return fit, res_ptr, scale_arr, scale_ptr

    def step(self, rays, img_gts, prefetch):
        """-> (loss tensor [view into the step's workspace], num_samples), or None when this batch has to go through the
        Python-issued step (buffers too small, empty batch, stale bf16 copy): nothing has been changed then."""
        C = _hip()
        t, d = self.t, self.t._direct
        if not self._ensure(rays):
            return None
        lib, h = C.lib, self.handle
        dev = rays.origins.device
        stream = C._stream()
        blas = self.blas
        # ---- the batch itself: counted one step ahead, or now
        p, self.pending = self.pending, None
        if p is None or p["rays"] is not rays:
            st, d._pending = d._pending, None
            # (counted ahead by the Python-issued step that took the previous batch: its seed, so that the jitter stream is one)
            seed = st["seed"] if (st is not None and st.get("rays") is rays) else blas._draw_seed()
            slot = self.next_slot
        elif p["slot"] is None or p["blas"] is not blas:
            seed, slot = p["seed"], self.next_slot                    # the octree was pruned since: redo it (same jitter stream)
        else:
            seed, slot = p["seed"], p["slot"]
            p = "counted"
        if p != "counted":
            C._check(lib.wisp_nerf_step_count(h, slot, rays.origins.data_ptr(), rays.dirs.data_ptr(), rays.origins.shape[0], seed, stream),
                     "nerf_step_count")
        self.next_slot = slot ^ 1
        gts = img_gts if (img_gts.dtype == torch.float32 and img_gts.is_contiguous()) else img_gts.float().contiguous()
        nxt = None
        if prefetch is not None and self.applies(prefetch, None) and float(prefetch.dist_min) == float(rays.dist_min) \
                and float(prefetch.dist_max) == float(rays.dist_max):
            nxt = dict(rays=prefetch, slot=slot ^ 1, seed=blas._draw_seed(), blas=blas)
        guess = t.pipeline.tracer.get_prev_num_samples() or t.target_sample_size
        fit, res_ptr, scale_arr, scale_ptr = self._hashgrid_scratch(dev, int(guess))
        hp = self._hyper()
        S = ctypes.c_int64(0)
        cov = (ctypes.c_int64 * 16)()
        loss_ptr = ctypes.c_void_p()
        record = 1 if C.TIMING is not None else 0
        t.wait_for_parameters()
        rc = lib.wisp_nerf_step_run(h, slot, gts.data_ptr(),
                                    None if nxt is None else nxt["rays"].origins.data_ptr(), None if nxt is None else nxt["rays"].dirs.data_ptr(),
                                    0 if nxt is None else nxt["rays"].origins.shape[0], 0 if nxt is None else nxt["seed"],
                                    ctypes.byref(hp), scale_ptr, self.hg_ws.data_ptr(), self.hg_ws.numel(), record,
                                    ctypes.byref(S), ctypes.cast(cov, ctypes.c_void_p), ctypes.byref(loss_ptr), stream)
        if rc == C.WISP_ERR_CAPACITY:
            # more samples than the buffers hold (or none): the Python-issued step takes this batch, counted again from the same seed
            self.fallbacks += 1
            d._pending = d._count(rays, None, seed=seed)
            d._next_seed = None if nxt is None else nxt["seed"]
            return None
        C._check(rc, "nerf_step_run")
        self.pending = nxt
        self.steps += 1
        n = int(S.value)
        # bookkeeping the Python step does around its launches
        t.pipeline.tracer.prev_num_samples = n
        t.opt_steps += 1
        if hp.optimizer == 2:
            L = len(d.res)
            first = d._first_idx_host
            cover = [min(int(cov[l]), first[l + 1] - first[l]) for l in range(L)]
            ga, gb = t.flat.ranges["grid"]
            t.fused_elements_last = 2 * sum(max(0, c) for c in cover)
        else:
            t.fused_elements_last = 0
        t.flat.mark_shadow_current()
        if fit is not None:
            ws_bytes = self.hg_ws.numel()
            if fit.pending is None and fit.calls % fit.CHECK_EVERY == 0:
                ws_bytes = int(C.lib.wisp_hashgrid_bwd_workspace_bytes(n, 3, C.BF16, 2, res_ptr, len(d.res), d.bitwidth, scale_ptr))
            fit.after_launch(n, res_ptr, scale_arr, scale_ptr, self.hg_ws, ws_bytes)
        addr = loss_ptr.value
        view = self.loss_view.get(addr)
        if view is None:
            ws = self.keep[0]
            o = addr - ws.data_ptr()
            view = self.loss_view[addr] = ws[o:o + 4].view(torch.float32)
        return view, n

    def drain_timing(self, sink):
        """move the step's own HIP-event record (wisp_nerf_step_read_timing; synchronise first) into bench.py's timing sink"""
        if self.handle is None or sink is None:
            return
        C = _hip()
        cap = 2048
        ms = (ctypes.c_float * (4 * cap))()
        units = (ctypes.c_int64 * cap)()
        n = ctypes.c_int32(0)
        C._check(C.lib.wisp_nerf_step_read_timing(self.handle, cap, ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(units, ctypes.c_void_p),
                                                  ctypes.byref(n)), "nerf_step_read_timing")
        for i in range(int(n.value)):
            for k, name in enumerate(self.TIMED):
                sink.setdefault(name, []).append((_Elapsed(float(ms[4 * i + k])), None, int(units[i])))
