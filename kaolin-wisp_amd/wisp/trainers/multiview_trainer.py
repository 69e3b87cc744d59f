"""MultiviewTrainStep: the optimisation step of the reference's MultiviewTrainer, without its app plumbing.

Semantics follow wisp/trainers/multiview_trainer.py:85-180 (pre_step pruning, warm-up raymarch, loss over rays,
adaptive ray count) and wisp/trainers/base_trainer.py:205-246 (parameter groups: names containing 'decoder' get
weight decay, names containing 'grid' get lr * grid_lr_weight; MultiStepLR).  MI355X-specific choices:
  * all trainable parameters live in ONE flat fp32 buffer (views keep their reference names), so the optimizer is a
    single fused AdamW launch per group (csrc/misc.hip) that also zeroes the gradients, and
  * data-parallel training is one RCCL all-reduce of the flat gradient buffer per step (the reference has no
    distributed path at all); rays are sharded across ranks, parameters and the occupancy octree are replicated;
  * bf16 autocast replaces the reference's fp16 autocast + GradScaler (no loss scaling needed).
"""
import logging as log
import math
import os
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from wisp.core import Rays
from wisp.ops.grid import current_shadow, mark_shadow_current, register_table_aux
from wisp.trainers.base_trainer import BaseTrainer, ConfigBaseTrainer


def _hip():
    import wisp._C as _C
    return _C


# profiler ranges at the sites the reference marks (torch.cuda.nvtx == roctx on ROCm; a no-op without a profiler attached)
def _range_push(msg):
    torch.cuda.nvtx.range_push(msg)


def _range_pop():
    torch.cuda.nvtx.range_pop()


class FlatParams:
    """Re-homes the trainable parameters of `module` into one flat buffer, grouped like init_optimizer does."""

    def __init__(self, module: torch.nn.Module):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        groups = {"decoder": [], "grid": [], "rest": []}
        for n, p in named:
            groups["decoder" if "decoder" in n else "grid" if "grid" in n else "rest"].append((n, p))
        device = named[0][1].device
        total = sum(((p.numel() + 3) // 4) * 4 for _, p in named)          # 16-byte aligned segments
        self.data = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.ranges = {}
        off = 0
        for g in ("decoder", "grid", "rest"):
            start = off
            for n, p in groups[g]:
                k = p.numel()
                self.data[off:off + k].copy_(p.detach().reshape(-1).float())
                p.data = self.data[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                if "grid" in n:
                    register_table_aux(p, grad_buffer=p.grad)   # hash-grid backward scatters straight into the flat buffer
                off += ((k + 3) // 4) * 4
            self.ranges[g] = (start, off)
        self.exp_avg = torch.zeros_like(self.data)
        self.exp_avg_sq = torch.zeros_like(self.data)
        self.shadow = None
        self._grid_params = [(p, (p.data_ptr() - self.data.data_ptr()) // 4) for n, p in groups["grid"]]

    def enable_bf16_shadow(self):
        """Keep a bf16 copy of the grid parameters, refreshed by the fused AdamW kernel, and hand it to the hash-grid op
        (`wisp.ops.grid.register_table_aux`), so that the bf16 forward needs no cast pass over the 41.7 MB table."""
        a, b = self.ranges["grid"]
        if b <= a:
            return
        self.shadow = self.data[a:b].to(torch.bfloat16)
        for p, off in self._grid_params:
            register_table_aux(p, shadow=self.shadow[off - a:off - a + p.numel()].view(p.shape))

    def refresh_shadow(self):
        """Re-derive the shadow from the master weights where a torch-side write (load_state_dict, ...) outdated it."""
        if self.shadow is None:
            return
        a, _ = self.ranges["grid"]
        for p, off in self._grid_params:
            if current_shadow(p, self.shadow.dtype) is None:
                self.shadow[off - a:off - a + p.numel()].copy_(p.detach().reshape(-1))
                mark_shadow_current(p)

    def mark_shadow_current(self):
        for p, _ in self._grid_params:
            mark_shadow_current(p)


# WISP_FUSED_COMPOSITE=0: the direct-issue step composites, takes the loss and back-propagates through the compositing with the
# three separate launches of the modular path instead of wisp_composite_loss
FUSED_COMPOSITE_LOSS = os.environ.get("WISP_FUSED_COMPOSITE", "1") != "0"


class _DirectNeRFStep:
    """The forward + loss + backward of MultiviewTrainStep.step for NeuralRadianceField + PackedRFTracer pipelines over an
    OctreeAS, issued as the same HIP launches in the same order as Pipeline.forward + autograd would issue them - minus the
    nn.Module / channel-negotiation / RenderBuffer / autograd-engine plumbing, which costs more host time per step than the
    GPU needs for a 2^18-sample batch (and left the GPU idle a quarter of the time in the 'voxel' configurations).

    Two tiers:
      * everything but the grid is always issued directly: raymarch (its parameter-free half - the occupancy count of 'ray',
        the cell intersection count of 'voxel' / 'uniform' - one batch ahead when the caller names the next batch), fused
        decoder forward / backward, compositing + loss + compositing backward (one launch: wisp_composite_loss);
      * the grid lookup is issued directly for the nerf_hash.yaml shape ('cat' HashGrid whose table gradient lives in the flat
        buffer) and for multi-level trilinear OctreeGrid / CodebookOctreeGrid fields (nerf_octree.yaml, nerf_codebook.yaml: one
        launch for all levels forward, gradients written straight into the parameters' .grad); any other grid of the plugin
        surface (TriplanarGrid, 'sum' HashGrid, one-level or 'closest' octree grids) keeps its own `interpolate` - one small
        autograd graph whose backward is seeded with the decoder's input gradient.
    Results are identical to the modular path (tests/test_gpu_1_selfcheck.py::test_direct_step_equals_modular_step)."""

    @staticmethod
    def supports(pipeline):
        from wisp.accelstructs import OctreeAS
        from wisp.models.nefs import NeuralRadianceField
        from wisp.tracers import PackedRFTracer
        from wisp.ops.nerf_mlp import SUPPORTED, FUSED_HIDDEN
        nef, tracer = pipeline.nef, pipeline.tracer
        if type(nef) is not NeuralRadianceField or type(tracer) is not PackedRFTracer:
            return False
        grid = nef.grid
        if grid is None or type(getattr(grid, 'blas', None)) is not OctreeAS or grid.blas.max_level > 10:
            return False
        if getattr(tracer, 'raymarch_type', None) not in ('ray', 'voxel', 'uniform'):
            return False
        return (nef.fused_decoder and nef.hidden_dim in FUSED_HIDDEN and nef.num_layers == 1             # (hidden 128: bf16 only, see __init__)
                and nef.view_multires == SUPPORTED["view_freqs"] and nef.pos_embedder is None
                and nef.view_embedder_type == 'positional' and nef.activation_type == 'relu'
                and nef.layer_type in ('linear', 'none') and 1 <= nef.effective_feature_dim() <= SUPPORTED["max_in_dim"]
                and getattr(nef, 'decoder_compute', 'auto') == 'auto')

    def __init__(self, trainer):
        from wisp.models.grids import HashGrid
        from wisp.ops.nerf_mlp import _flat_view, _pack, SUPPORTED, BF16_ONLY_HIDDEN
        self.t = trainer
        nef = trainer.pipeline.nef
        grid = nef.grid
        H, I = nef.hidden_dim, nef.effective_feature_dim()
        self.shape = (I, H, SUPPORTED["view_freqs"])
        self.param_shapes = ((H, I), (H,), (16, H), (16,), (H, 42), (H,), (H, H), (H,), (3, H), (3,))
        dec = _decoder_tensors(nef)
        self.dec = dec
        self.biasless = any(t is None for t in dec)                          # nerf_codebook.yaml: decoders without bias
        if not self.biasless:
            self.packed = _flat_view([p.detach() for p in dec])              # zero-copy views of the flat buffers
            self.packed_grad = _flat_view([p.grad for p in dec])
            self.ok = self.packed is not None and self.packed_grad is not None
        else:
            # absent biases are zero rows of the packed parameter vector; their gradient slots are scratch
            self.packed = self.packed_grad = None
            self._scratch_grad = torch.zeros(sum(int(np.prod(sh)) for sh in self.param_shapes), dtype=torch.float32,
                                             device=trainer.flat.data.device)
            self.ok = all(p is None or p.grad is not None for p in dec)
        if H in BF16_ONLY_HIDDEN and not trainer.enable_amp:
            self.ok = False                                                  # the wide decoder kernels compute in bf16 only
        self.hash_fast = type(grid) is HashGrid and grid.multiscale_type == 'cat' and grid.codebook.feats.grad is not None
        if self.hash_fast:
            cb = grid.codebook
            self.table = cb.feats
            self.first_idx = cb.begin_idxes
            self.res = [int(r) for r in cb.resolutions.reshape(-1).tolist()]
            self.bitwidth = grid.codebook_bitwidth
            # the tracer queries lod_idx = num_lods - 1 and 'cat' zeroes the columns from lod_idx * feature_dim on
            # (reference hash_grid.py:226-229): the finest level's columns are zero, exactly as in the modular path
            self.zero_from_col = (grid.num_lods - 1) * grid.feature_dim
            self._first_idx_host = [int(v) for v in cb.begin_idxes.detach().cpu().reshape(-1).tolist()]
        self.octree_tier = None if self.hash_fast else self._octree_tier(grid)
        self._pending = None
        self._cells = (None, None, 1)
        if self.biasless:
            # persistent packed parameter vector: bias slots stay zero, the weight segments are refreshed by one multi-tensor copy
            self._packed = torch.zeros_like(self._scratch_grad)
            off, self._w_dst, self._w_src, self._g_dst, self._g_src = 0, [], [], [], []
            for prm, sh in zip(dec, self.param_shapes):
                n = int(np.prod(sh))
                if prm is not None:
                    self._w_dst.append(self._packed[off:off + n].view(sh))
                    self._w_src.append(prm.detach())
                    self._g_dst.append(prm.grad)
                    self._g_src.append(self._scratch_grad[off:off + n].view(sh))
                off += n
            self.ok = self.ok and all(a.shape == b.shape and b.dtype == torch.float32 for a, b in zip(self._w_dst, self._w_src))

    # ---- octree / codebook feature grids ---------------------------------------------------------------------------------
    @staticmethod
    def _octree_tier(grid):
        """'octree' | 'codebook' | None: which multi-level trilinear field the direct lookup below covers."""
        from wisp.models.grids import CodebookOctreeGrid, OctreeGrid
        if type(grid) not in (OctreeGrid, CodebookOctreeGrid) or grid.interpolation_type != 'linear' or grid.num_lods < 2 \
                or grid.num_lods > 16 or grid.multiscale_type not in ('cat', 'sum'):
            return None
        prm = list(grid.features) + (list(grid.dictionary) if type(grid) is CodebookOctreeGrid else [])
        if not all(q.is_cuda and q.dtype == torch.float32 and q.ndim == 2 and q.grad is not None and q.grad.is_contiguous()
                   and q.grad.dtype == torch.float32 for q in prm):
            return None                                  # frozen / baked / half tables: the grid's own interpolate handles them
        if type(grid) is OctreeGrid:
            return 'octree' if grid._fusable() else None
        K, F = grid.dictionary[0].shape
        return 'codebook' if (grid.fused and F <= K <= 256 and F <= 16) else None      # (the fused kernels' shapes)

    def _octree_forward(self, samples):
        """OctreeGrid.interpolate(samples, num_lods - 1) (octree_grid.py:183-219) / CodebookOctreeGrid's (codebook_grid.py:
        103-172) without the autograd graph: cell chain query, (codebook: every logits row decoded once), all levels' trilinear
        blends in one launch.  -> features f32 [S, F or L*F], context for _octree_backward."""
        C = _hip()
        grid = self.t.pipeline.nef.grid
        L = grid.num_lods
        grid._sync_device(samples.device)
        tr = grid.trinkets if grid.trinkets.dtype == torch.int32 else grid.trinkets.int()
        # the cell chain of the active levels; the 'voxel' march already knows every sample's cell at the level it marched
        # (octree_grid.py:221-226: base_lod), which spares the walk down to there
        blas = grid.blas
        hint, hint_level, group = self._cells
        if hint is None or hint_level != grid.base_lod or hint.shape[0] * group != samples.shape[0]:
            hint, group = None, 1
        chain = blas.query_chain(samples, grid.active_lods[L - 1], grid.base_lod, hint, group)
        levels = grid.active_lods[:L]
        summed = grid.multiscale_type == 'sum'
        if self.octree_tier == 'codebook':
            tables = [C.codebook_decode_rows(grid.features[i], grid.dictionary[i], True) for i in range(L)]
            half = False
        else:
            tables = [grid.features[i].detach() for i in range(L)]
            half = grid.half_features
        feats = C.spc_trilinear_multi_forward(samples, chain, grid.blas.points, tr, tables, levels, half, summed)
        return feats, (samples, chain, tr, levels, summed)

    def _octree_backward(self, ctx, g_feats):
        """Feature-table gradients, written straight into the parameters' .grad (zero since the last optimizer step)."""
        C = _hip()
        grid = self.t.pipeline.nef.grid
        samples, chain, tr, levels, summed = ctx
        L, F = grid.num_lods, grid.feature_dim
        if g_feats.dtype != torch.float32:
            g_feats = g_feats.float()
        if self.octree_tier == 'octree':
            C.spc_trilinear_multi_backward(samples, chain, grid.blas.points, tr, g_feats, [tuple(f.shape) for f in grid.features[:L]],
                                           levels, summed, out=[f.grad for f in grid.features[:L]])
            return
        C.codebook_trilinear_multi_backward(samples, chain, grid.blas.points, tr, [grid.features[i].detach() for i in range(L)],
                                            [grid.dictionary[i].detach() for i in range(L)], g_feats, levels, summed,
                                            out=([grid.features[i].grad for i in range(L)], [grid.dictionary[i].grad for i in range(L)]))

    # ---- decoder parameters ------------------------------------------------------------------------------------------
    def _params(self):
        if not self.biasless:
            return self.packed, self.packed_grad
        self._scratch_grad.zero_()
        torch._foreach_copy_(self._w_dst, self._w_src)
        return self._packed, self._scratch_grad

    def _scatter_param_grads(self, g):
        """bias-free decoders: the kernel wrote one packed gradient vector; add its weight segments to the parameters' .grad"""
        assert g is self._scratch_grad
        torch._foreach_add_(self._g_dst, self._g_src)

    # ---- raymarch ------------------------------------------------------------------------------------------------------
    def _count(self, rays, jitter, seed=None):
        C = _hip()
        pipe = self.t.pipeline
        blas = pipe.nef.grid.blas
        if torch.is_tensor(rays.dist_min) or torch.is_tensor(rays.dist_max):
            raise TypeError("'ray' raymarch needs scalar Rays.dist_min / dist_max (as the reference, octree_as.py:276-277)")
        blas._to_device(rays.origins.device)
        level = blas.max_level                             # where HashGrid.raymarch marches (hash_grid.py:235-240)
        coarse, lc = blas._coarse_bitfield(rays, pipe.tracer.num_steps, level)
        st = C.raymarch_ray_count(blas._bitfield(level), blas.octree, blas.prefix, rays.origins, rays.dirs, rays.dist_min,
                                  rays.dist_max, pipe.tracer.num_steps, level, jitter,
                                  blas._draw_seed() if seed is None else seed, coarse, lc)
        st["rays"], st["blas"] = rays, blas
        return st

    def _march(self, rays, jitter, prefetch, coded):
        """-> ridx, samples, deltas, per-ray sample offsets, per-sample view directions (None when `coded`)"""
        C = _hip()
        pipe = self.t.pipeline
        tracer, grid = pipe.tracer, pipe.nef.grid
        blas = grid.blas
        from wisp.models.grids import HashGrid
        if tracer.raymarch_type == 'ray' and type(grid) is HashGrid and blas._bitfield(blas.max_level) is not None:
            # (only the hash grid marches at the octree's finest level whatever the lod; every other grid picks its own level
            #  inside grid.raymarch - octree_grid.py:221-226 marches at base_lod)
            st, self._pending = self._pending, None
            if st is None or st["rays"] is not rays or jitter is not None:
                st = self._count(rays, jitter)                # nothing was prefetched for this batch
            elif st["blas"] is not blas:
                st = self._count(rays, None, seed=st["seed"])  # the octree was pruned since: redo it (same jitter stream)
            if coded:
                ridx, samples, depths, deltas, boundary, offsets = C.raymarch_ray_finish(st)
                dirs = None
            else:
                ridx, samples, depths, deltas, boundary, offsets, dirs = C.raymarch_ray_finish(st, with_dirs=True)
            if prefetch is not None:
                # (a seed the native step already drew for this batch before it handed the step over: one jitter stream either way)
                seed, self._next_seed = getattr(self, "_next_seed", None), None
                self._pending = self._count(prefetch, None, seed=seed)
            return ridx, samples, deltas, offsets, dirs
        st, self._pending = self._pending, None
        kw = {} if (jitter is None or tracer.raymarch_type == 'uniform') else {"jitter": jitter}
        lvl = grid.active_lods[grid.num_lods - 1]
        ahead = tracer.raymarch_type in ('voxel', 'uniform')            # (supports(): the blas is an OctreeAS)
        if ahead and isinstance(st, dict) and st.get("rays") is rays:
            kw["begun"] = st                              # (checked against octree and level again where it is used)
        rm = grid.raymarch(rays, level=lvl, num_samples=tracer.num_steps, raymarch_type=tracer.raymarch_type, **kw)
        if ahead and prefetch is not None:
            # the next batch's cell intersection counts, behind this step's own march: its size read-back is long done when the
            # next step asks for it, so the host never waits for the GPU to drain
            self._pending = grid.raymarch(prefetch, level=lvl, num_samples=tracer.num_steps, raymarch_type=tracer.raymarch_type,
                                          begin_only=True)
        dirs = None if coded else rays.dirs.index_select(0, rm.ridx)
        self._cells = (getattr(rm, "nugget_pidx", None), getattr(rm, "nugget_level", None), getattr(rm, "samples_per_nugget", 1))
        return rm.ridx, rm.samples, rm.deltas, rm.ray_offsets, dirs

    def run(self, rays, img_gts, jitter=None, prefetch=None, fused_update=None):
        """-> (loss tensor, num_samples); gradients are left accumulated in the parameters' .grad (the flat gradient buffer).
        `fused_update` (MultiviewTrainStep._fused_update_args; hash-grid tier only): the table's AdamW step is folded into the
        grid backward for the rows whose reduce workgroup owns them - their gradient never reaches the flat buffer; which rows
        those were is left in trainer._fused_cover for the optimizer step that follows to skip.  A caller that passes
        `fused_update` MUST call trainer.reduce_and_update() right after: those rows (parameter, moments, bf16 copy) have taken
        step `opt_steps + 1` already and only that call advances opt_steps and steps the remaining rows.
        `prefetch`: the Rays of the NEXT step; with the 'ray' march their parameter-free prefix (occupancy test + offsets) is
        issued now, behind this step's own raymarch, so that the next step finds its sample count already computed instead of
        stalling the GPU on the size read-back."""
        C = _hip()
        t = self.t
        pipe = t.pipeline
        nef, tracer = pipe.nef, pipe.tracer
        grid = nef.grid
        dev = rays.origins.device
        N = rays.origins.shape[0]
        # view directions: encoded once per ray and gathered by ray index inside the decoder kernels where that variant exists
        # (the training shape), else gathered per sample
        i, h, f = self.shape
        coded = C.nerf_mlp_rays_preferred(torch.bfloat16 if t.enable_amp else torch.float32, i, h, f, t.enable_amp)
        ridx, samples, deltas, offsets, dirs = self._march(rays, jitter, prefetch, coded)
        S = samples.shape[0]
        tracer.prev_num_samples = S
        t.wait_for_parameters()                 # everything above overlapped the previous step's all-reduce + update
        if self.hash_fast:
            table = self.table
            if t.enable_amp:
                shadow = current_shadow(table, torch.bfloat16)
                table = shadow if shadow is not None else table.to(torch.bfloat16)
            feats = C.hashgrid_interpolate(samples, table.detach(), self.first_idx, self.res, self.bitwidth, self.zero_from_col)
            feats_in = feats
        elif self.octree_tier is not None and grid.training:
            feats, octx = self._octree_forward(samples)
            feats_in = feats
        else:
            octx = None
            # any other grid: its own interpolate (one small autograd graph), evaluated like Pipeline.forward would
            with torch.enable_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=t.enable_amp):
                feats = grid.interpolate(samples, grid.num_lods - 1).reshape(S, i)
            feats_in = feats.detach()
            if not feats_in.is_contiguous():
                feats_in = feats_in.contiguous()
        packed, packed_grad = self._params()
        ray_code = (ridx, C.nerf_mlp_dir_code(rays.dirs, f)) if coded else None
        color, density = C.nerf_mlp_forward(feats_in, dirs, packed, i, h, f, t.enable_amp, ray_code=ray_code)
        if tracer.bg_color.device != dev:
            tracer.bg_color = tracer.bg_color.to(dev)
        bg = tracer._bg_host()
        if t.rgb_loss_type not in ('huber', 'l2', 'l1'):
            raise NotImplementedError
        if FUSED_COMPOSITE_LOSS:
            # compositing, the loss with its gradient w.r.t. the composited colours (what autograd derives for
            # loss_fn(...).mean()) and the compositing backward: one launch, one pass per ray
            loss, g_color, g_density, _ = C.composite_loss(color, density, deltas, offsets, N, bg, img_gts, t.rgb_loss_type)
            loss = loss[0]
        else:
            rgb, _alpha, _depth, _hit, _w = C.composite_fwd(color, density, deltas, None, None, offsets, N, bg)
            loss, g_rgb = C.rgb_loss(rgb, img_gts, t.rgb_loss_type)
            loss = loss[0]
            g_color, g_density = C.composite_bwd(g_rgb, None, None, color, density, deltas, None, None, offsets, bg)
        _range_push("MultiviewTrainer.backward")
        try:
            g_feats, _ = C.nerf_mlp_backward(feats_in, dirs, packed, g_color, g_density, i, h, f, t.enable_amp,
                                             grad_params=packed_grad, ray_code=ray_code)
            if self.biasless:
                self._scatter_param_grads(packed_grad)
            t.early_reduce_decoder()                # decoder gradients are final: their all-reduce runs under the grid backward
            if self.hash_fast:
                if fused_update is not None:
                    _, t._fused_cover = C.hashgrid_interpolate_backward(samples, g_feats, tuple(self.table.shape), self.first_idx, self.res,
                                                                        self.bitwidth, self.zero_from_col, out=self.table.grad,
                                                                        adamw=fused_update)
                else:
                    C.hashgrid_interpolate_backward(samples, g_feats, tuple(self.table.shape), self.first_idx, self.res, self.bitwidth,
                                                    self.zero_from_col, out=self.table.grad)
            elif self.octree_tier is not None and grid.training:
                self._octree_backward(octx, g_feats)
            elif feats.requires_grad:
                feats.backward(g_feats.to(feats.dtype))
        finally:
            _range_pop()
        return loss, S


def _decoder_tensors(nef):
    from wisp.ops.nerf_mlp import _decoder_tensors as dt
    return dt(nef)


class _StaleMasterGuard:
    """state_dict pre-hook of the sharded optimizer.  Holds the trainer weakly and pickles to an inert object, so that a pipeline
    saved whole (`torch.save(pipeline)`, the reference's 'full' format) does not drag the trainer along."""

    def __init__(self, trainer):
        self._trainer = weakref.ref(trainer)

    def __call__(self, module, prefix, keep_vars):
        trainer = self._trainer()
        if trainer is not None and trainer._master_stale:
            raise RuntimeError("sharded optimizer: the fp32 table rows owned by other ranks are stale on this rank (only their bf16 "
                               "shadow travelled); call sync_master() on EVERY rank before state_dict() / saving a checkpoint")

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self._trainer = lambda: None


class MultiviewTrainStep:
    def __init__(self, pipeline, lr=1e-3, eps=1e-16, weight_decay=1e-6, grid_lr_weight=500.0, betas=(0.9, 0.999),
                 rgb_loss_type='huber', prune_every=100, target_sample_size=2 ** 18, max_rays=2 ** 18,
                 enable_amp=False, scheduler_milestones=None, scheduler_gamma=0.333, process_group=None, seed=0,
                 optimizer='adamw', alpha=0.99, momentum=0.0, prune_rng_device=None, sharded_optimizer=None):
        """optimizer: 'adamw' | 'adam' | 'rmsprop' - the torch.optim classes the reference's configs select
        (wisp/config/presets/torch.py:45-68; nerf_hash.yaml: AdamW, nerf_octree / nerf_codebook.yaml: RMSprop), each
        one fused launch over the flat parameter buffer.  `betas` (Adam family) / `alpha`, `momentum` (RMSprop) as in torch."""
        self.pipeline = pipeline
        self.optimizer = str(optimizer).lower()
        if self.optimizer not in ('adamw', 'adam', 'rmsprop'):
            raise ValueError(f"optimizer must be 'adamw', 'adam' or 'rmsprop', got {optimizer!r}")
        self.alpha, self.momentum = alpha, momentum
        self.flat = FlatParams(pipeline.nef)
        self.lr, self.eps, self.weight_decay, self.grid_lr_weight, self.betas = lr, eps, weight_decay, grid_lr_weight, betas
        self.rgb_loss_type = rgb_loss_type
        self.prune_every = prune_every
        self.target_sample_size = target_sample_size
        self.max_rays = max_rays
        self.enable_amp = enable_amp
        if enable_amp:
            self.flat.enable_bf16_shadow()
        self.milestones = sorted(scheduler_milestones or [])
        self.gamma = scheduler_gamma
        self.total_iterations = 0
        self.opt_steps = 0
        self.num_rays = None
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.group = process_group
        # WISP_FORCE_ALLREDUCE=1 runs the collective even with one rank (exercises the RCCL path on a single-GPU box)
        self.force_allreduce = (os.environ.get("WISP_FORCE_ALLREDUCE", "0") == "1" and dist.is_available()
                                and dist.is_initialized())
        # prune draws must be identical on every rank so the replicated octrees stay identical: every rank seeds the same
        # generator.  By default it lives on the parameters' device (the 2 x [cells, 3] draws of a level-7 prune take ~60 ms
        # on the host plus 50 MB of H2D, against microseconds on the GPU); prune_rng_device='cpu' reproduces host draws
        # (what the CPU oracle can replay: scripts/psnr_parity.py).
        dev = self.flat.data.device if prune_rng_device is None else torch.device(prune_rng_device)
        self._prune_gen = torch.Generator(device=dev).manual_seed(seed)
        self._side_stream = None
        self._params_ready = None
        # Gradient exchange with more than one rank (see _sharded_reduce_and_update): reduce-scatter + optimizer on this rank's
        # slice of the grid + all-gather of the bf16 shadow, instead of all-reduce + replicated optimizer - 25 % fewer bytes on
        # the wire (4 + 2 per element against 4 + 4) and the optimizer on 1/world of the table.  It is the default when there IS
        # a shadow to gather (amp) and more than one rank; without amp the fp32 master would have to travel back and nothing is
        # saved, so the all-reduce stays.  sharded_optimizer=True / False or WISP_SHARDED_OPTIM=1 / 0 force either path.
        env = os.environ.get("WISP_SHARDED_OPTIM", "")
        if sharded_optimizer is not None:
            want = bool(sharded_optimizer)
        elif env in ("0", "1"):
            want = env == "1"
        else:
            want = bool(enable_amp) and self.world > 1
        self.sharded_optimizer = bool(want and dist.is_available() and dist.is_initialized())
        self._decoder_reduced = False
        self.comm_timing = None           # a list: reduce_and_update() appends (events...) per step on GPU runs (bench.py, world > 1)
        self.rank = dist.get_rank(process_group) if self.sharded_optimizer else 0
        self._master_stale = False
        self._stage = {}
        self._plan = None
        if self.sharded_optimizer:
            # a state_dict taken while other ranks' slices of the fp32 table are stale would be a silently wrong checkpoint; the
            # hook cannot run the collective itself (a checkpoint is usually written by one rank), so it refuses instead
            pipeline.nef.register_state_dict_pre_hook(_StaleMasterGuard(self))
        # (single GPU: putting the optimizer launch on the side stream too, under the next step's raymarch, was measured
        # neutral - 1.270 vs 1.284 ms/step - so one rank keeps everything on one stream)
        # specialised issue order for the flagship pipeline shape (WISP_DIRECT_STEP=0 keeps the modular path)
        self._direct = None
        self._last_step_modular = True
        # One GPU, AdamW, the nerf_hash.yaml table: the grid's optimizer step is folded into the hash-grid backward's reduce kernel
        # (see _fused_update_args).  WISP_ADAM_IN_FLUSH=0 / fuse_grid_optimizer = False keep the separate pass.
        self.fuse_grid_optimizer = os.environ.get("WISP_ADAM_IN_FLUSH", "1") != "0"
        self._fused_cover = None
        self.fused_elements_last = 0
        # gradient accumulation: accumulate() runs forward + backward of a micro-batch and leaves its gradient ADDED in the flat
        # buffer; the step() that follows exchanges and applies the MEAN over grad_accum_steps micro-batches - the arithmetic of
        # grad_accum_steps data-parallel ranks on one GPU (each rank's loss is a mean over its own rays, the all-reduced sum is
        # divided by the world size).  scripts/time_to_psnr.py emulates the 8-GPU weak-scaling batch with it.
        self.grad_accum_steps = 1
        self._native = None
        if os.environ.get("WISP_DIRECT_STEP", "1") != "0" and _DirectNeRFStep.supports(pipeline):
            d = _DirectNeRFStep(self)
            self._direct = d if d.ok else None
            if self._direct is not None and self._direct.hash_fast:
                # the whole step issued by one call into the library where its shape is covered (csrc/train_step.hip); decided per batch
                from wisp.trainers._native_step import NativeHashStep
                self._native = NativeHashStep(self)

    # -------------------------------------------------------------------------------------------- schedule / groups
    def _lr_scale(self, step=None):
        step = self.opt_steps if step is None else step
        k = sum(1 for m in self.milestones if step >= m)                 # MultiStepLR
        return self.gamma ** k

    def _fused_update_args(self):
        """Arguments of C.hashgrid_interpolate_backward(adamw=...) for the step that is about to run, or None when the grid's
        optimizer step cannot be folded into its backward: more than one rank (the gradient is exchanged first), another
        optimizer, another grid tier.  The update is optimizer_step's own for the grid group - same learning rate schedule,
        same step count, same arithmetic (wisp_adamw_update) - applied by the reduce workgroup that owns a table slice instead
        of by a second pass: 12 of the 42 bytes per parameter the two passes move stay in LDS, and the rest of the optimizer's
        traffic runs under the reduce kernel's record walk."""
        d, f = self._direct, self.flat
        if (not self.fuse_grid_optimizer or d is None or not d.hash_fast or self.optimizer != 'adamw' or self.world > 1
                or self.force_allreduce or not f.data.is_cuda or d.table.shape[1] != 2 or getattr(self, "grad_accum_steps", 1) > 1):
            return None
        if 'optimizer_step' in self.__dict__ or type(self).optimizer_step is not MultiviewTrainStep.optimizer_step \
                or type(self).reduce_and_update is not MultiviewTrainStep.reduce_and_update:
            return None                         # someone replaced the optimizer step (instance or subclass): it gets the whole gradient
        off = next((o for p, o in f._grid_params if p is d.table), None)
        if off is None:
            return None
        n = d.table.numel()
        a, _ = f.ranges["grid"]
        step = self.opt_steps + 1
        return dict(param=d.table.detach(), exp_avg=f.exp_avg[off:off + n].view(d.table.shape),
                    exp_avg_sq=f.exp_avg_sq[off:off + n].view(d.table.shape),
                    shadow=None if f.shadow is None else f.shadow[off - a:off - a + n].view(d.table.shape),
                    lr=self.lr * self.grid_lr_weight * self._lr_scale(step), beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                    weight_decay=self.weight_decay, step=step, grad_scale=1.0 / self.world)

    def _uncovered_grid_ranges(self, covered):
        """[lo, hi) element ranges of the flat buffer inside the grid group that the fused update did NOT touch."""
        d, f = self._direct, self.flat
        a, b = f.ranges["grid"]
        off = next(o for p, o in f._grid_params if p is d.table)
        F, first = d.table.shape[1], d._first_idx_host
        out, at = [], a
        for l, rows in enumerate(covered):
            rows = min(int(rows), first[l + 1] - first[l])
            if rows <= 0:
                continue
            lo, hi = off + first[l] * F, off + (first[l] + rows) * F
            if lo > at:
                out.append((at, lo))
            at = max(at, hi)
        if b > at:
            out.append((at, b))
        self.fused_elements_last = (b - a) - sum(hi - lo for lo, hi in out)      # (bench.py: the optimizer's share of the launch)
        return out

    def optimizer_step(self, grid_ranges=None):
        """grid_ranges: None -> the whole grid group; else a list of [lo, hi) element ranges of the flat buffer inside the grid
        group (the sharded path: this rank's slice, plus the rows no gradient can reach)."""
        C = _hip()
        self.opt_steps += 1
        s = self._lr_scale()
        f = self.flat
        gs = 1.0 / (self.world * max(1, int(getattr(self, "grad_accum_steps", 1))))
        groups = []
        for g, lr in (("decoder", self.lr), ("grid", self.lr * self.grid_lr_weight), ("rest", self.lr)):
            a, b = f.ranges[g]
            if b <= a:
                continue
            if g != "grid" or grid_ranges is None:
                groups.append((a, b - a, lr * s, self.weight_decay, f.shadow[:b - a] if (g == "grid" and f.shadow is not None) else None))
                continue
            for lo, hi in grid_ranges:
                if hi > lo:
                    groups.append((lo, hi - lo, lr * s, self.weight_decay, None if f.shadow is None else f.shadow[lo - a:hi - a]))
        # all parameter groups in ONE launch (the decoder group alone is ~10 K parameters)
        if self.optimizer == 'adamw':
            C.adamw_step_groups(f.data, f.grad, f.exp_avg, f.exp_avg_sq, groups, self.betas[0], self.betas[1], self.eps,
                                self.opt_steps, grad_scale=gs, zero_grad=True)
        elif self.optimizer == 'adam':
            C.optim_step_groups('adam', f.data, f.grad, f.exp_avg, f.exp_avg_sq, groups, self.betas[0], self.betas[1],
                                self.eps, self.opt_steps, grad_scale=gs, zero_grad=True)
        else:       # RMSprop: exp_avg_sq holds square_avg, exp_avg the momentum buffer
            C.optim_step_groups('rmsprop', f.data, f.grad, f.exp_avg if self.momentum > 0 else None, f.exp_avg_sq, groups,
                                self.alpha, self.momentum, self.eps, self.opt_steps, grad_scale=gs, zero_grad=True)
        f.mark_shadow_current()                 # the kernel rewrote every shadow element from the new master weights

    def _live_grad_numel(self):
        """How much of the flat gradient buffer can be non-zero after a step of the direct-issue path: the tracer queries
        lod_idx = num_lods - 1 and 'cat' zeroes the columns of the FINEST level (reference hash_grid.py:226-229), so that level's
        rows - the tail of the buffer when the table is its last tensor - never receive a gradient and need not travel:
        4 MB of the 41.8 MB of nerf_hash.yaml.  (Their weight decay is the same arithmetic on every rank.)"""
        f, d = self.flat, getattr(self, "_direct", None)
        if d is None or not d.hash_fast or getattr(self, "_last_step_modular", True):
            return f.grad.numel()
        a, b = f.ranges["grid"]
        ra, rb = f.ranges["rest"]
        if rb > ra or len(f._grid_params) != 1 or f._grid_params[0][0] is not d.table:
            return f.grad.numel()
        F = d.table.shape[1]
        live_levels = min((d.zero_from_col + F - 1) // F, len(d.res))
        if live_levels >= len(d.res):
            return f.grad.numel()
        first = getattr(d, "_first_idx_host", None)
        if first is None:
            first = d._first_idx_host = [int(v) for v in d.first_idx.detach().cpu().reshape(-1).tolist()]
        return f._grid_params[0][1] + first[live_levels] * F

    def early_reduce_decoder(self):
        """The decoder group's gradients are final as soon as the decoder backward has run - 0.4 ms (the hash-grid backward)
        before the table's.  Their all-reduce (41 KB: pure latency) is issued right then on the side stream, so it is out of
        the way when the grid's collective starts.  Called by the direct-issue step; the modular path reduces everything at the
        end.  Sums are the same numbers as in the late reduction."""
        if not (self.world > 1 or self.force_allreduce) or self._decoder_reduced:
            return
        a, b = self.flat.ranges["decoder"]
        if b <= a:
            return
        if self.flat.data.is_cuda:
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream()
            self._side_stream.wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=self.group)
        self._decoder_reduced = True

    def allreduce_grads(self):
        if self.world > 1 or self.force_allreduce:
            if getattr(self, "_decoder_reduced", False):
                # the decoder group went ahead (early_reduce_decoder): reduce what lies behind it
                self._decoder_reduced = False
                a = self.flat.ranges["decoder"][1]
                n = max(self._live_grad_numel(), a)
                if n > a:
                    dist.all_reduce(self.flat.grad[a:n], op=dist.ReduceOp.SUM, group=self.group)
                return
            n = self._live_grad_numel()
            if n < self.flat.grad.numel() and os.environ.get("WISP_CHECK_GRAD_TAIL", "0") == "1":
                # debug switch (a host sync per step): the rows left out of the collective must not have received a gradient
                assert float(self.flat.grad[n:].abs().max()) == 0.0, "a gradient reached table rows the all-reduce skips"
            dist.all_reduce(self.flat.grad[:n], op=dist.ReduceOp.SUM, group=self.group)     # RCCL over xGMI

    def reduce_and_update(self):
        """Gradient all-reduce + optimizer.  With more than one rank both run on a side stream, so that the part of the
        NEXT step that does not read parameters (ray gathering, raymarch against the occupancy structure, its size
        read-back) overlaps the collective; whoever reads parameters next calls wait_for_parameters() first."""
        if not (self.world > 1 or self.force_allreduce) or not self.flat.data.is_cuda:
            self._reduce_and_update_here()
            return
        main = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        self._side_stream.wait_stream(main)
        timing = self.comm_timing
        with torch.cuda.stream(self._side_stream):
            if timing is not None:
                start = torch.cuda.Event(enable_timing=True)
                start.record()
                self._timing_marks = []            # (label, event) after every phase of _reduce_and_update_here
            self._reduce_and_update_here()
            self._params_ready = torch.cuda.Event(enable_timing=timing is not None)
            self._params_ready.record()
            if timing is not None:
                marks, self._timing_marks = self._timing_marks, None
                timing.append(dict(start=start, marks=marks, done=self._params_ready, waited_at=None))

    def _mark(self, label):
        marks = getattr(self, "_timing_marks", None)
        if marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((label, ev))

    def _reduce_and_update_here(self):
        plan = self._shard_plan() if (self.world > 1 or self.force_allreduce) else None
        if plan is None:
            self.allreduce_grads()
            self._mark("all_reduce")
            cover, self._fused_cover = self._fused_cover, None
            if cover is not None and any(cover):
                self.optimizer_step(self._uncovered_grid_ranges(cover))
            else:
                self.fused_elements_last = 0
                self.optimizer_step()
            self._mark("optimizer")
        else:
            self._sharded_reduce_and_update(plan)

    # -------------------------------------------------------------------------------------------- sharded optimizer (opt-in)
    def _shard_plan(self):
        """How the grid group is cut over the ranks, or None when this step goes through the all-reduce.
        Only the elements a gradient can reach ([ga, ga + live), see _live_grad_numel) are cut, into `world` slices of `c`
        elements (c a multiple of 4 = the optimizer kernel's alignment).  direct: the padded window [ga, ga + world*c) lies inside
        the grid group, so the collectives run on views of the flat buffers; otherwise (the pad would run into the next group:
        small or oddly sized tables) they go through zero-padded staging buffers."""
        if not self.sharded_optimizer:
            return None
        f = self.flat
        ga, gb = f.ranges["grid"]
        live_end = min(self._live_grad_numel(), gb)
        if self._plan is not None:
            # The cut is fixed by the first sharded step: the optimizer state of a slice lives on its owner only, and the rows
            # behind the window are updated by every rank on the assumption that no gradient reaches them.
            if self._plan and live_end > min(self._plan["ga"] + self._plan["npad"], gb):
                raise RuntimeError("sharded optimizer: this step's gradient reaches table rows outside the partition fixed at the "
                                   "first step (the step path changed mid-run); use the all-reduce path (WISP_SHARDED_OPTIM=0)")
            return self._plan or None
        live = live_end - ga
        if live < 4 * self.world:
            self._plan = False                            # table too small to cut: all-reduce for the whole run
            return None
        c = ((live + 4 * self.world - 1) // (4 * self.world)) * 4
        npad = c * self.world
        lo = ga + self.rank * c
        self._plan = dict(ga=ga, gb=gb, c=c, npad=npad, direct=ga + npad <= gb, lo=min(lo, gb), hi=min(lo + c, gb))
        return self._plan

    def _staging(self, name, numel, dtype):
        t = self._stage.get(name)
        if t is None or t.numel() != numel or t.dtype != dtype:
            t = self._stage[name] = torch.zeros(numel, dtype=dtype, device=self.flat.data.device)
        return t

    def _gather_grid(self, full, plan, tag):
        """all-gather of every rank's slice of `full` (the grid group's master weights or their bf16 shadow, [gb - ga] elements).
        The send buffer is always a copy, never a view of the receive buffer."""
        ga, gb, c, npad, lo, hi = (plan[k] for k in ("ga", "gb", "c", "npad", "lo", "hi"))
        own = self._staging("own_" + tag, c, full.dtype)
        own[:hi - lo].copy_(full[lo - ga:hi - ga])
        out = full[:npad] if plan["direct"] else self._staging("all_" + tag, npad, full.dtype)
        dist.all_gather_into_tensor(out, own, group=self.group)
        if not plan["direct"]:
            full[:gb - ga].copy_(out[:gb - ga])

    def _sharded_reduce_and_update(self, plan):
        """Reduce-scatter + optimizer on this rank's slice + all-gather, for the grid group only; the decoder
        (and any other small group) keeps its all-reduce and replicated update.  Against all-reduce + replicated optimizer:
        the optimizer touches 1/world of the table, and with the bf16 shadow the wire carries 4 + 2 bytes per element
        instead of 4 + 4 (only the shadow travels back; the fp32 master of the other ranks' slices goes STALE until
        sync_master(), which prune() calls - everything else on the training path reads the shadow).  Without a shadow the
        fp32 master is gathered every step and never stale.  Optimizer state (exp_avg, exp_avg_sq) exists only for the own slice.
        The sums are the same numbers in the same order as the all-reduce's for 2 ranks; for more, RCCL's ring order applies
        to both."""
        f = self.flat
        ga, gb, c, npad, lo, hi = (plan[k] for k in ("ga", "gb", "c", "npad", "lo", "hi"))
        early, self._decoder_reduced = self._decoder_reduced, False
        for name in ("decoder", "rest"):
            a, b = f.ranges[name]
            if b > a and not (name == "decoder" and early):
                dist.all_reduce(f.grad[a:b], op=dist.ReduceOp.SUM, group=self.group)
        if plan["direct"]:
            send = f.grad[ga:ga + npad]
        else:
            send = self._staging("rs_in", npad, torch.float32)
            send[:gb - ga].copy_(f.grad[ga:gb])
        mine = self._staging("rs_out", c, torch.float32)
        dist.reduce_scatter_tensor(mine, send, op=dist.ReduceOp.SUM, group=self.group)
        f.grad[ga:min(ga + npad, gb)].zero_()             # the other ranks' slices of this rank's gradient are spent
        f.grad[lo:hi].copy_(mine[:hi - lo])
        self._mark("reduce_scatter")
        self.optimizer_step(grid_ranges=[(lo, hi), (min(ga + npad, gb), gb)])
        self._mark("optimizer")
        if f.shadow is not None:
            self._gather_grid(f.shadow, plan, "bf16")
            self._master_stale = self.world > 1
        else:
            self._gather_grid(f.data[ga:gb], plan, "fp32")
        self._mark("all_gather")

    def _refuse_stale_master_forward(self):
        """While other ranks' slices of the fp32 table are stale here, the forward may only read the bf16 shadow.  If a torch-side write
        (load_state_dict, an external optimizer, ...) has outdated the shadow, the ops would fall back to casting the master - stale
        rows, silently.  Refuse instead."""
        if not self._master_stale:
            return
        for p, _ in self.flat._grid_params:
            if current_shadow(p, self.flat.shadow.dtype) is None:
                raise RuntimeError("sharded optimizer: the bf16 shadow of the table was outdated by a write outside the trainer while the "
                                   "fp32 master of other ranks' slices is stale; call sync_master() on every rank, then "
                                   "flat.refresh_shadow(), before the next step")

    def sync_master(self):
        """Collective (every rank): bring the fp32 master weights of the other ranks' slices up to date after sharded steps
        that only exchanged the bf16 shadow.  Needed before anything reads the table in fp32: prune() (does it itself),
        checkpoints, fp32 evaluation."""
        if not self._master_stale:
            return
        self.wait_for_parameters()
        plan = self._plan
        f = self.flat
        self._gather_grid(f.data[plan["ga"]:plan["gb"]], plan, "fp32")
        f.mark_shadow_current()          # the shadow already holds bf16(master) everywhere; the in-place copy bumped versions
        self._master_stale = False

    def wait_for_parameters(self):
        """Order the current stream after the last reduce_and_update()."""
        ev = self._params_ready
        if ev is not None:
            if self.comm_timing:
                here = torch.cuda.Event(enable_timing=True)        # where the main stream stood when it had to have the parameters
                here.record()
                self.comm_timing[-1]["waited_at"] = here
            torch.cuda.current_stream().wait_event(ev)
            self._params_ready = None

    # -------------------------------------------------------------------------------------------- reference hooks
    def comm_summary(self):
        """Per-step milliseconds of the recorded steps (call after a synchronize): collectives, optimizer, and how long the main
        stream really stood waiting for the parameters (what the overlap did not hide)."""
        rows = [r for r in (self.comm_timing or []) if r["waited_at"] is not None]
        if not rows:
            return None
        n = len(rows)
        phases = {}
        for r in rows:
            prev = r["start"]
            for label, ev in r["marks"]:
                phases[label] = phases.get(label, 0.0) + prev.elapsed_time(ev)
                prev = ev
        total = sum(r["start"].elapsed_time(r["done"]) for r in rows) / n
        exposed = sum(max(0.0, r["waited_at"].elapsed_time(r["done"])) for r in rows) / n
        out = dict(steps=n, side_stream_ms=total, exposed_wait_ms=exposed, hidden_ms=max(0.0, total - exposed),
                   path="reduce-scatter + sharded optimizer + all-gather" if self.sharded_optimizer else "all-reduce + replicated optimizer")
        for label, v in phases.items():
            out[label + "_ms"] = v / n
        out["collective_ms"] = sum(v for k, v in phases.items() if k != "optimizer") / n
        return out

    def pre_step(self):
        """multiview_trainer.py:85-93."""
        if self.prune_every > -1 and self.total_iterations > 1 and self.total_iterations % self.prune_every == 0:
            self.prune()

    def prune(self):
        """nef.prune() with rank-identical draws.  Like the reference (nerf.py:181-183: prune is a no-op unless the grid is
        a HashGrid with both prune densities set), grids without an occupancy record are left alone."""
        self.wait_for_parameters()
        nef = self.pipeline.nef
        if (getattr(nef, 'prune_density_decay', None) is None or getattr(nef, 'prune_min_density', None) is None
                or getattr(nef.grid, 'dense_points', None) is None or not hasattr(nef, 'prune')):
            return
        self.sync_master()                      # the density query below runs in fp32 on the master weights
        cells = nef.grid.dense_points.shape[0]
        dev = self._prune_gen.device
        unit = torch.rand(cells, 3, generator=self._prune_gen, device=dev)
        views = torch.nn.functional.normalize(torch.randn(cells, 3, generator=self._prune_gen, device=dev), dim=1)
        nef.prune(unit_samples=unit, view_dirs=views)

    def calc_adaptive_rays(self, num_rays_in_batch):
        """multiview_trainer.py:95-109: rays for the next step so that ~target_sample_size samples are produced."""
        spr = self.pipeline.tracer.get_prev_num_samples() / max(num_rays_in_batch, 1)
        n = self.target_sample_size / max(spr, 1)
        self.num_rays = int(math.floor(min(n, self.max_rays)))
        return self.num_rays

    def loss_fn(self, rgb, gts):
        if self.rgb_loss_type == 'l2':
            return torch.nn.functional.mse_loss(rgb, gts, reduction='none').mean()
        if self.rgb_loss_type == 'l1':
            return torch.abs(rgb - gts).mean()
        if self.rgb_loss_type == 'huber':
            return torch.nn.functional.smooth_l1_loss(rgb, gts, reduction='none').mean()
        raise NotImplementedError

    def native_timing_into(self, sink):
        """bench.py: the native step brackets its four roofline entry points with its own HIP events; after a synchronize this
        moves them into the (start, end, units) sink the Python-issued launches fill through wisp._C.TIMING."""
        if self._native is not None:
            self._native.drain_timing(sink)

    def accumulate(self, rays: Rays, img_gts, jitter=None):
        """Forward + backward of ONE micro-batch: its gradient is ADDED to the flat buffer, nothing is exchanged or applied.
        Call grad_accum_steps - 1 times, then step() with the last micro-batch (set grad_accum_steps first: it is the divisor).
        Returns (loss tensor, num_samples)."""
        assert self.grad_accum_steps > 1, "set grad_accum_steps to the number of micro-batches per optimizer step first"
        self._refuse_stale_master_forward()
        if self._direct is not None and self.pipeline.nef.training:
            with torch.no_grad():
                loss, _ = self._direct.run(rays, img_gts, jitter, None, fused_update=None)
        else:
            self.wait_for_parameters()
            kw = {} if jitter is None else {"jitter": jitter}
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=self.enable_amp):
                rb = self.pipeline(rays=rays, lod_idx=None, channels=["rgb"], **kw)
                loss = self.loss_fn(rb.rgb.float(), img_gts)
            loss.backward()
        return loss.detach(), self.pipeline.tracer.get_prev_num_samples()

    def step(self, rays: Rays, img_gts, jitter=None, prefetch: Optional[Rays] = None):
        """One optimisation step on this rank's ray shard.  Returns (loss tensor, num_samples).
        `prefetch` (optional): the Rays object the NEXT call will be given - the direct-issue path then runs their
        occupancy test early (a data-loader style look-ahead; results are the same with or without it)."""
        _range_push("MultiviewTrainer.step")                 # the reference's profiler ranges (multiview_trainer.py:111,169)
        try:
            self.pre_step()
            self._refuse_stale_master_forward()
            self.total_iterations += 1
            self._fused_cover = None                         # (a step that raised between run() and optimizer_step leaves none behind)
            self._last_step_modular = not (self._direct is not None and self.pipeline.nef.training)
            done = None
            if not self._last_step_modular and self._native is not None and self._native.applies(rays, jitter):
                # one call into the library issues every launch of the step, the optimizer included (csrc/train_step.hip)
                done = self._native.step(rays, img_gts, prefetch)
            if done is not None:
                loss = done[0]
                self.calc_adaptive_rays(rays.origins.shape[0])
                return loss, done[1]
            if not self._last_step_modular:
                with torch.no_grad():
                    loss, _ = self._direct.run(rays, img_gts, jitter, prefetch, fused_update=self._fused_update_args())
            else:
                self.wait_for_parameters()
                kw = {} if jitter is None else {"jitter": jitter}
                with torch.autocast('cuda', dtype=torch.bfloat16, enabled=self.enable_amp):
                    rb = self.pipeline(rays=rays, lod_idx=None, channels=["rgb"], **kw)
                    loss = self.loss_fn(rb.rgb.float(), img_gts)
                _range_push("MultiviewTrainer.backward")
                try:
                    loss.backward()
                finally:
                    _range_pop()
            self.reduce_and_update()
            self.calc_adaptive_rays(rays.origins.shape[0])
        finally:
            _range_pop()
        return loss.detach(), self.pipeline.tracer.get_prev_num_samples()


@dataclass
class ConfigMultiviewTrainer(ConfigBaseTrainer):
    """Field names and defaults of wisp/trainers/multiview_trainer.py:33-63 (the YAML schema under `trainer:`)."""
    start_prune: int = 1000
    prune_every: int = 100
    random_lod: bool = False
    rgb_lambda: float = 1.0
    opacity_loss: float = 0.0
    rgb_loss_type: str = 'l2'
    rgb_loss_denom: str = 'rays'
    target_sample_size: int = 2 ** 18
    save_valid_imgs: bool = False


class MultiviewTrainer(BaseTrainer):
    """The reference's multiview trainer as an application uses it (wisp/trainers/multiview_trainer.py:65-180; constructed
    like app/nerf/main_nerf.py:110): the *unchanged-trainer* regime over this package's Pipeline - autograd through the
    modular HIP ops, fp16 autocast + GradScaler, a torch.optim optimizer with the reference's parameter groups, two
    `.item()` read-backs per step for the loss metrics.  Same events in the same order as the reference's step(); the fused
    MI355X step that replaces all of this is MultiviewTrainStep (same losses, same parameters afterwards; tests/).
    Validation renders through wisp.trainers.validation (PSNR only; lpips / ssim need packages that are out of scope)."""

    def __init__(self, cfg, pipeline, train_dataset, validation_dataset=None, tracker=None, device='cuda', scene_state=None):
        super().__init__(cfg=cfg, pipeline=pipeline, train_dataset=train_dataset, tracker=tracker, device=device,
                         scene_state=scene_state)
        self.validation_dataset = validation_dataset

    def pre_step(self):
        super().pre_step()
        every = self.cfg.prune_every
        if every > -1 and self.total_iterations > 1 and self.total_iterations % every == 0:
            self.pipeline.nef.prune()

    def calc_adaptive_rays(self, rays, warmup=False):
        pipe = self.pipeline
        if warmup:
            marched = pipe.nef.grid.raymarch(rays, level=pipe.nef.grid.active_lods[-1], num_samples=pipe.tracer.num_steps,
                                             raymarch_type=pipe.tracer.raymarch_type)
            pipe.tracer.prev_num_samples = marched.samples.shape[0]
        per_ray = pipe.tracer.get_prev_num_samples() / rays.shape[0]
        num_rays = int(math.floor(min(self.cfg.target_sample_size / max(per_ray, 1), 2 ** 18)))
        transform = getattr(self.train_dataset, 'transform', None)
        if not hasattr(transform, 'set_num_samples') or type(transform).__name__ != 'SampleRays':
            raise Exception("SampleRays should be used as the transform for the dataset")
        transform.set_num_samples(num_rays)

    def step(self, data):
        # the reference's profiler range (multiview_trainer.py:111: @torch.cuda.nvtx.range("MultiviewTrainer.step")) as a push / pop
        # pair - the decorator form costs a contextlib round trip per call in a host-bound loop
        _range_push("MultiviewTrainer.step")
        try:
            return self._step(data)
        finally:
            _range_pop()

    def _step(self, data):
        rays = data['rays'].to(self.device).squeeze(0)
        img_gts = data['rgb'].to(self.device).squeeze(0)
        tracer = self.pipeline.tracer
        if tracer.get_prev_num_samples() is None:
            self.calc_adaptive_rays(rays, warmup=True)           # first call only sizes the batch, no optimisation
            return
        self.optimizer.zero_grad()
        lod_idx = None
        if self.cfg.random_lod:
            import random
            lods = self.pipeline.nef.grid.num_lods
            total = float(sum(2 ** i for i in range(lods)))
            lod_idx = random.choices(list(range(lods)), [2 ** i / total for i in range(lods)])[0]
        rb = self.pipeline(rays=rays, lod_idx=lod_idx, channels=["rgb"])
        kind = self.cfg.rgb_loss_type
        if kind == 'l2':
            per_elem = torch.nn.functional.mse_loss(rb.rgb, img_gts, reduction='none')
        elif kind == 'l1':
            per_elem = torch.abs(rb.rgb - img_gts)
        elif kind == 'huber':
            per_elem = torch.nn.functional.smooth_l1_loss(rb.rgb, img_gts, reduction='none')
        else:
            raise NotImplementedError
        denom = self.cfg.rgb_loss_denom
        if denom == 'samples':
            rgb_loss = per_elem.sum() / tracer.prev_num_samples
        elif denom == 'rays':
            rgb_loss = per_elem.mean()
        else:
            raise NotImplementedError
        loss = 0 + rgb_loss
        if self.cfg.opacity_loss > 0.0 and self.total_iterations < 1000:
            loss = loss + self.cfg.opacity_loss * ((1.0 - rb.alpha) ** 2).mean()
        m = self.tracker.metrics
        m.total_loss += loss.item()
        m.rgb_loss += rgb_loss.item()
        m.num_samples += 1
        _range_push("MultiviewTrainer.backward")                        # (multiview_trainer.py:169-177)
        try:
            if self.cfg.enable_amp:
                self.scaler.scale(loss).backward()
                self.scaler.step(self.optimizer)
                self.scaler.update()
            else:
                loss.backward()
                self.optimizer.step()
        finally:
            _range_pop()
        self.calc_adaptive_rays(rays, warmup=False)
        if self.cfg.scheduler:
            self.scheduler.step()

    def log_console(self):
        m = self.tracker.metrics
        log.info('EPOCH {}/{} | total loss: {:>.3E} | rgb loss: {:>.3E}'.format(
            self.epoch, self.max_epochs, m.average_metric('total_loss'), m.average_metric('rgb_loss')))

    def validate(self):
        """PSNR over the validation views (multiview_trainer.py:257-303 without image / table writers)."""
        if self.validation_dataset is None:
            return None
        from wisp.trainers.validation import evaluate_psnr
        data = self.validation_dataset.data
        rays, rgb = data["rays"], data["rgb"]
        views = [(rays[i], rgb[i]) for i in range(rays.origins.shape[0])]
        mean, line = evaluate_psnr(self.pipeline, views, epoch=self.epoch, max_epochs=self.max_epochs)
        log.info(line)
        self.return_dict = {"psnr": mean}
        return self.return_dict


def shard_rays(num_rays: int, rank: int, world: int):
    """Contiguous, disjoint, exhaustive ray shards: rank r gets rays [lo, hi).  Rays are independent (each ray's samples,
    field queries and compositing touch no other ray), so this is the only data-path partitioning the method needs."""
    base, rem = divmod(num_rays, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
