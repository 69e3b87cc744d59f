"""PackedSDFTracer: sphere tracing of a neural SDF through the occupied cells of its octree (NGLOD rendering).
Surface and stepping rule of wisp/tracers/packed_sdf_tracer.py:21-175; the octree raytrace, the first-hit marking and
find_depth_bound are HIP kernels (csrc/spc.hip, csrc/render.hip).  The per-iteration field query still goes through the
neural field's own forward (a persistent fused sphere-tracing kernel is the planned next step for this row)."""
import torch
import torch.nn.functional as F

from wisp.core import RenderBuffer
from wisp.ops.differential import finitediff_gradient
from wisp.ops.geometric import find_depth_bound
from wisp.tracers.base_tracer import BaseTracer
import wisp.ops.render as render_ops


class PackedSDFTracer(BaseTracer):
    def __init__(self, num_steps=1024, step_size=0.8, min_dis=0.0003):
        """num_steps: max sphere-tracing iterations; step_size: scale of every step; min_dis: convergence distance."""
        super().__init__()
        self.num_steps = num_steps
        self.step_size = step_size
        self.min_dis = min_dis

    def get_supported_channels(self):
        return {"depth", "normal", "xyz", "hit", "rgb", "alpha"}

    def get_required_nef_channels(self):
        return {"sdf"}

    def trace(self, nef, rays, channels, extra_channels, lod_idx=None, num_steps=64, step_size=1.0, min_dis=1e-4):
        """Sphere-trace `rays`; returns RenderBuffer(xyz, depth, hit, normal, rgb (= normal colours), alpha)."""
        assert nef.grid is not None and "this tracer requires a grid"
        if lod_idx is None:
            lod_idx = nef.grid.num_lods - 1
        invres = 1.0
        rt = nef.grid.raytrace(rays, nef.grid.active_lods[lod_idx], with_exit=True)
        ridx, pidx, depth = rt.ridx, rt.pidx, rt.depth
        depth[..., 0:1] += 1e-5
        first_hit = render_ops.mark_pack_boundaries(ridx)
        curr_idxes = torch.nonzero(first_hit)[..., 0].int()
        first_ridx = ridx[first_hit].long()
        nug_o, nug_d = rays.origins[first_ridx], rays.dirs[first_ridx]
        mask = torch.ones([first_ridx.shape[0]], device=nug_o.device).bool()
        hit = torch.zeros_like(mask).bool()
        t = depth[first_hit][..., 0:1]
        x = torch.addcmul(nug_o, nug_d, t)
        dist = torch.zeros_like(t)
        curr_pidx = pidx[first_hit].long()

        def query(points, which):
            return nef(coords=points, lod_idx=lod_idx, pidx=curr_pidx[which], channels="sdf") * invres * step_size

        with torch.no_grad():
            if mask.any():
                dist[mask] = query(x[mask], mask).to(dist.dtype)
            dist[~mask] = 20
            dist_prev = dist.clone()
            for _ in range(num_steps):
                t += dist
                mcol = mask.view(-1, 1)
                x = torch.where(mcol, torch.addcmul(nug_o, nug_d, t), x)
                hit = torch.where(mask, torch.abs(dist)[..., 0] < min_dis * invres, hit)
                hit |= torch.where(mask, torch.abs(dist + dist_prev)[..., 0] * 0.5 < (min_dis * 5) * invres, hit)
                mask = torch.where(mask, (t < rays.dist_max)[..., 0], mask)
                mask &= ~hit
                if not mask.any():
                    break
                dist_prev = torch.where(mask.view(-1, 1), dist, dist_prev)
                next_idxes = find_depth_bound(t, depth, first_hit, curr_idxes=curr_idxes)
                mask &= (next_idxes != -1)
                aabb_mask = (next_idxes != curr_idxes)
                curr_idxes = torch.where(mask, next_idxes, curr_idxes)
                t = torch.where((mask & aabb_mask).view(-1, 1), depth[curr_idxes.long(), 0:1], t)
                x = torch.where(mask.view(-1, 1), torch.addcmul(nug_o, nug_d, t), x)
                curr_pidx = torch.where(mask, pidx[curr_idxes.long()].long(), curr_pidx)
                if not mask.any():
                    break
                dist[mask] = query(x[mask], mask).to(dist.dtype)

        o = rays.origins
        x_buffer = torch.zeros_like(o)
        depth_buffer = torch.zeros_like(o[..., 0:1])
        hit_buffer = torch.zeros_like(o[..., 0]).bool()
        normal_buffer = torch.zeros_like(o)
        rgb_buffer = torch.zeros(*o.shape[:-1], 3, device=o.device)
        alpha_buffer = torch.zeros(*o.shape[:-1], 1, device=o.device)
        hit_buffer[first_ridx] = hit
        extra_outputs = {}
        for channel in extra_channels:
            feats = nef(coords=x[hit], lod_idx=lod_idx, channels=channel)
            buf = torch.zeros(*o.shape[:-1], feats.shape[-1], device=feats.device)
            buf[hit_buffer] = feats.to(buf.dtype)
            extra_outputs[channel] = buf
        x_buffer[hit_buffer] = x[hit]
        depth_buffer[hit_buffer] = t[hit]
        if "rgb" in channels or "normal" in channels:
            if hit.any():
                grad = finitediff_gradient(x[hit], nef.get_forward_function("sdf"))
                normal_buffer[hit_buffer] = F.normalize(grad, p=2, dim=-1, eps=1e-5)
            rgb_buffer[..., :3] = (normal_buffer + 1.0) / 2.0
        alpha_buffer[hit_buffer] = 1.0
        return RenderBuffer(xyz=x_buffer, depth=depth_buffer, hit=hit_buffer, normal=normal_buffer, rgb=rgb_buffer,
                            alpha=alpha_buffer, **extra_outputs)
