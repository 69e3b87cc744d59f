"""Oracle (test infrastructure): packed (ragged) volume integration in torch-CPU with autograd.

Restates the Kaolin-Core 0.13 leaves kaolin.render.spc.{cumsum, sum_reduce, exponential_integration}
as used at wisp/tracers/packed_rf_tracer.py:143-165 (semantics: SURVEY.md Appendix A.4/A.5), and the
tracer's compositing block itself.  Pure torch ops, so autograd provides the backward the HIP kernels
are checked against.  Run in float64 for a tight reference, float32 for the timed CPU baseline.

Parity: the leaves (cumsum, sum_reduce, exponential_integration) are UNPINNED restatements of Kaolin's published semantics, checked
against a float64 python loop; the compositing block built on them is PINNED to PackedRFTracer.trace compiled from the reference file.
"""
import torch


def pack_ids(boundary):
    """pack index of every sample: inclusive count of boundary flags minus one."""
    return torch.cumsum(boundary.to(torch.int64), 0) - 1


def sum_reduce(feats, boundary):
    """kaolin sum_reduce (packed_rf_tracer.py:157,160): segmented sum -> [P, C], P = boundary.sum()."""
    P = int(boundary.sum())
    out = torch.zeros(P, feats.shape[1], dtype=feats.dtype)
    if feats.shape[0]:
        out = out.index_add(0, pack_ids(boundary), feats)
    return out


def cumsum(feats, boundary, exclusive=False):
    """kaolin cumsum: running sum restarted at every pack boundary."""
    if feats.shape[0] == 0:
        return feats
    inc = torch.cumsum(feats, 0)
    ids = pack_ids(boundary)
    starts = torch.nonzero(boundary)[:, 0]
    before = torch.cat([torch.zeros(1, feats.shape[1], dtype=feats.dtype), inc[:-1]], 0)[starts]   # total before each pack
    inc = inc - before[ids]
    return inc - feats if exclusive else inc


def exponential_integration(feats, tau, boundary, exclusive=True):
    """kaolin exponential_integration (packed_rf_tracer.py:154): alpha = 1-exp(-tau);
    T = exp(-cumsum(tau, exclusive)); w = T*alpha; returns (sum_reduce(w*feats), w)."""
    alpha = 1.0 - torch.exp(-tau)
    T = torch.exp(-cumsum(tau, boundary, exclusive=exclusive))
    w = T * alpha
    return sum_reduce(w * feats, boundary), w


def composite(color, density, deltas, depths, ridx, boundary, num_rays, bg_color, with_depth=True):
    """The compositing block of PackedRFTracer.trace (packed_rf_tracer.py:143-165).
    Returns dict(rgb [R,3], alpha [R,1], depth [R,1] or None, hit bool [R])."""
    dt = color.dtype
    S = color.shape[0]
    bg = torch.as_tensor(bg_color, dtype=dt)
    rgb = torch.zeros(num_rays, 3, dtype=dt) + bg
    hit = torch.zeros(num_rays, dtype=torch.bool)
    out_alpha = torch.zeros(num_rays, 1, dtype=dt)
    depth = torch.zeros(num_rays, 1, dtype=dt) if with_depth else None
    ridx_hit = ridx[boundary]
    tau = density.reshape(S, 1) * deltas.to(dt)
    ray_colors, w = exponential_integration(color, tau, boundary, exclusive=True)
    if with_depth:
        depth = depth.index_put((ridx_hit,), sum_reduce(depths.reshape(S, 1).to(dt) * w, boundary))
    alpha = sum_reduce(w, boundary)
    out_alpha = out_alpha.index_put((ridx_hit,), alpha)
    hit[ridx_hit] = alpha[..., 0].detach() > 0.0
    rgb = rgb.index_put((ridx_hit,), bg * (1.0 - alpha) + ray_colors)
    return dict(rgb=rgb, alpha=out_alpha, depth=depth, hit=hit, weights=w)
