"""PackedSDFTracer: sphere tracing of a neural SDF through the occupied cells of its octree (NGLOD rendering).

Same tracer surface and stepping rule as wisp/tracers/packed_sdf_tracer.py:21-175, organised differently: the marching
state lives in a small struct of flat device buffers, and everything a marching iteration does apart from the field query
(advance, convergence and far-plane tests, search of the next occupied cell, jump to its entry point, new query position)
is ONE HIP launch (`wisp_sphere_trace_step`, csrc/render.hip) instead of ~25 masked tensor ops.  The octree raytrace and
the first-hit marking are HIP kernels too (csrc/spc.hip).
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from wisp.core import RenderBuffer
from wisp.ops.differential import finitediff_gradient
from wisp.tracers.base_tracer import BaseTracer
import wisp.ops.render as render_ops


@dataclass
class _MarchState:
    """Per-pack (ray that owns at least one nugget) marching state; every tensor has leading dimension P."""
    ray: torch.Tensor          # int64  ray index of the pack
    o: torch.Tensor            # f32 [P,3] origin
    d: torch.Tensor            # f32 [P,3] direction
    t: torch.Tensor            # f32 [P]   current depth
    x: torch.Tensor            # f32 [P,3] current position
    dist: torch.Tensor         # f32 [P]   last (scaled) signed distance
    dist_prev: torch.Tensor    # f32 [P]
    active: torch.Tensor       # u8  [P]   still marching
    hit: torch.Tensor          # u8  [P]   converged on the surface
    nug: torch.Tensor          # i32 [P]   current nugget
    nug_next: torch.Tensor     # i32 [P]   scratch (double buffer)
    cell: torch.Tensor         # i64 [P]   point-hierarchy index of the current cell


class PackedSDFTracer(BaseTracer):
    def __init__(self, num_steps=1024, step_size=0.8, min_dis=0.0003):
        """num_steps: marching iterations at most; step_size: multiplier on every SDF step; min_dis: convergence distance."""
        super().__init__()
        self.num_steps = num_steps
        self.step_size = step_size
        self.min_dis = min_dis

    def get_supported_channels(self):
        return {"depth", "normal", "xyz", "hit", "rgb", "alpha"}

    def get_required_nef_channels(self):
        return {"sdf"}

    # ------------------------------------------------------------------ pieces of trace()
    @staticmethod
    def _start(rays, ridx, pidx, depth):
        first = render_ops.mark_pack_boundaries(ridx)
        nug = torch.nonzero(first)[..., 0].int()
        ray = ridx[first].long()
        o, d = rays.origins[ray].contiguous(), rays.dirs[ray].contiguous()
        t = depth[first][..., 0].contiguous()
        P = ray.shape[0]
        dev = o.device
        return _MarchState(ray=ray, o=o, d=d, t=t, x=torch.addcmul(o, d, t[:, None]), dist=torch.zeros(P, device=dev),
                           dist_prev=torch.zeros(P, device=dev), active=torch.ones(P, dtype=torch.uint8, device=dev),
                           hit=torch.zeros(P, dtype=torch.uint8, device=dev), nug=nug, nug_next=torch.empty_like(nug),
                           cell=pidx[first].long())

    @staticmethod
    def _query(nef, st, lod_idx, scale):
        """scaled SDF at the active positions, scattered into st.dist."""
        sel = st.active.bool()
        if bool(sel.any()):
            sdf = nef(coords=st.x[sel], lod_idx=lod_idx, pidx=st.cell[sel], channels="sdf") * scale
            st.dist[sel] = sdf.reshape(-1).to(st.dist.dtype)
        return sel

    # ------------------------------------------------------------------ fused iteration (field query inside the launch)
    @staticmethod
    def _fused_field(nef, lod_idx):
        """The tensors wisp_sdf_trace_step_fused needs, or None when the field is not the shape it is built for: a plain
        NeuralSDF (nglod_octree.yaml) - OctreeGrid with 16 'sum'-med feature channels, linear interpolation, raw position
        input without embedding, one hidden relu layer with bias.  Anything else marches through `nef(...)` per iteration."""
        from wisp.models.grids.octree_grid import OctreeGrid
        from wisp.models.nefs.neural_sdf import NeuralSDF
        import os
        if os.environ.get("WISP_SDF_FUSED", "1") == "0" or type(nef) is not NeuralSDF or type(nef.grid) is not OctreeGrid:
            return None
        g, dec = nef.grid, nef.decoder
        if (g.multiscale_type != 'sum' or g.interpolation_type != 'linear' or g.feature_dim != 16 or lod_idx < 1
                or not nef.position_input or not isinstance(nef.pos_embedder, torch.nn.Identity)
                or nef.activation_type != 'relu' or nef.num_layers != 1 or len(dec.layers) != 1 or dec.skip
                or type(dec.layers[0]) is not torch.nn.Linear or type(dec.lout) is not torch.nn.Linear
                or dec.layers[0].bias is None or dec.lout.bias is None or dec.lout.out_features != 1
                or dec.layers[0].out_features > 256 or not g.features[0].is_cuda):
            return None
        n = lod_idx + 1
        feats = [g.features[i].detach().contiguous() for i in range(n)]
        if any(f.dtype != feats[0].dtype or f.shape[1] != 16 for f in feats):
            return None
        g._sync_device(feats[0].device)
        return dict(feats=feats, levels=[int(l) for l in g.active_lods[:n]], half_round=bool(g.half_features),
                    w1=dec.layers[0].weight.detach().float().contiguous(), b1=dec.layers[0].bias.detach().float().contiguous(),
                    w2=dec.lout.weight.detach().float().reshape(-1).contiguous(), b2=dec.lout.bias.detach().float().contiguous(),
                    octree=g.blas.octree, exsum=g.blas.prefix, points=g.blas.points, trinkets=g.trinkets.int().contiguous())

    def _march_fused(self, fld, rays, rt_pidx, depth, st, num_steps, step_size, min_dis):
        """All marching iterations as one launch each, no host decision per iteration: the loop only peeks at a device
        counter of still-marching packs every 8 iterations (the reference reads `mask.any()` twice per iteration)."""
        import wisp._C as _C
        counter = torch.zeros(1, dtype=torch.int32, device=st.t.device)

        def launch(first, cnt):
            _C.sdf_trace_step_fused(first, st.o, st.d, depth, rt_pidx, rays.dist_max, min_dis, min_dis * 5, st.t, st.dist,
                                    st.dist_prev, st.active, st.hit, st.nug, st.nug_next, st.cell, st.x, fld["octree"],
                                    fld["exsum"], fld["points"], fld["trinkets"], fld["feats"], fld["levels"], fld["half_round"],
                                    fld["w1"], fld["b1"], fld["w2"], fld["b2"], step_size, cnt)
        launch(True, None)
        st.dist_prev.copy_(st.dist)
        for it in range(num_steps):
            peek = (it % 8) == 7
            if peek:
                counter.zero_()
            launch(False, counter if peek else None)
            st.nug, st.nug_next = st.nug_next, st.nug
            if peek and int(counter.item()) == 0:
                break

    def trace(self, nef, rays, channels, extra_channels, lod_idx=None, num_steps=64, step_size=1.0, min_dis=1e-4):
        """Sphere-trace `rays`; returns RenderBuffer(xyz, depth, hit, normal, rgb (= normal colours), alpha)."""
        import wisp._C as _C
        assert nef.grid is not None and "this tracer requires a grid"
        if lod_idx is None:
            lod_idx = nef.grid.num_lods - 1
        invres = 1.0
        rt = nef.grid.raytrace(rays, nef.grid.active_lods[lod_idx], with_exit=True)
        depth = rt.depth
        depth[..., 0:1] += 1e-5                               # start just inside the first cell
        st = self._start(rays, rt.ridx, rt.pidx, depth)
        fld = self._fused_field(nef, lod_idx) if st.t.shape[0] else None
        if fld is not None:
            with torch.no_grad():
                self._march_fused(fld, rays, rt.pidx, depth, st, num_steps, invres * step_size, min_dis * invres)
            return self._gather(nef, rays, st, channels, extra_channels, lod_idx)
        with torch.no_grad():
            self._query(nef, st, lod_idx, invres * step_size)
            st.dist_prev.copy_(st.dist)
            for _ in range(num_steps):
                _C.sphere_trace_step(st.o, st.d, depth, rt.pidx, rays.dist_max, min_dis * invres, (min_dis * 5) * invres,
                                     st.t, st.dist, st.dist_prev, st.active, st.hit, st.nug, st.nug_next, st.cell, st.x)
                st.nug, st.nug_next = st.nug_next, st.nug
                if not bool(self._query(nef, st, lod_idx, invres * step_size).any()):
                    break
        return self._gather(nef, rays, st, channels, extra_channels, lod_idx)

    @staticmethod
    def _gather(nef, rays, st, channels, extra_channels, lod_idx):
        """scatter the per-pack results into per-ray buffers (rays without nuggets keep zeros)."""
        o = rays.origins
        dev = o.device
        hit = st.hit.bool()
        out = dict(xyz=torch.zeros_like(o), depth=torch.zeros_like(o[..., 0:1]), hit=torch.zeros_like(o[..., 0]).bool(),
                   normal=torch.zeros_like(o), rgb=torch.zeros(*o.shape[:-1], 3, device=dev),
                   alpha=torch.zeros(*o.shape[:-1], 1, device=dev))
        out["hit"][st.ray] = hit
        on_surface = out["hit"]
        for channel in extra_channels:
            feats = nef(coords=st.x[hit], lod_idx=lod_idx, channels=channel)
            buf = torch.zeros(*o.shape[:-1], feats.shape[-1], device=feats.device)
            buf[on_surface] = feats.to(buf.dtype)
            out[channel] = buf
        out["xyz"][on_surface] = st.x[hit]
        out["depth"][on_surface] = st.t[hit][:, None]
        if "rgb" in channels or "normal" in channels:
            if bool(hit.any()):
                grad = finitediff_gradient(st.x[hit], nef.get_forward_function("sdf"))
                out["normal"][on_surface] = F.normalize(grad, p=2, dim=-1, eps=1e-5)
            out["rgb"][..., :3] = (out["normal"] + 1.0) / 2.0
        out["alpha"][on_surface] = 1.0
        return RenderBuffer(**out)
