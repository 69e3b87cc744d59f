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
