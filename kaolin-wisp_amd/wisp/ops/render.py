"""Packed volume-integration ops (autograd wrappers over csrc/render.hip).

Provide what the reference tracer takes from kaolin.render.spc - mark_pack_boundaries, cumsum, sum_reduce,
exponential_integration (wisp/tracers/packed_rf_tracer.py:154-173) - plus `composite`,
the fused replacement of the tracer's whole compositing block (:143-165).
"""
import torch


def _hip():
    import wisp._C as _C
    return _C


def mark_pack_boundaries(ids):
    """bool [n]: True where ids[i] != ids[i-1] (and at 0)."""
    return _hip().mark_pack_boundaries(ids)


def pack_starts(boundary):
    """int64 [P]: index of the first sample of every pack."""
    return _hip().pack_starts(boundary)


def pack_starts_of(raymarch_results):
    """Pack starts of an ASRaymarchResults; cached on the object as pack_info."""
    starts = raymarch_results.pack_info
    if starts is None:
        starts = pack_starts(raymarch_results.boundary)
        raymarch_results.pack_info = starts
    return starts


class _SumReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, starts):
        ctx.save_for_backward(starts)
        ctx.n = feats.shape[0]
        return _hip().packed_sum_reduce(feats.float(), starts)

    @staticmethod
    def backward(ctx, grad):
        (starts,) = ctx.saved_tensors
        # d/dfeats[i] = grad[pack(i)]: expand by pack
        n, P = ctx.n, starts.shape[0]
        lengths = torch.diff(starts, append=torch.tensor([n], device=starts.device, dtype=starts.dtype))
        return torch.repeat_interleave(grad, lengths, dim=0, output_size=n), None


def sum_reduce(feats, boundary, starts=None):
    """Segmented sum [S,C] -> [P,C] (deterministic wave reduction; the reference op uses atomics)."""
    if starts is None:
        starts = pack_starts(boundary)
    return _SumReduce.apply(feats, starts)


class _Cumsum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, starts, exclusive, reverse):
        ctx.save_for_backward(starts)
        ctx.flags = (exclusive, reverse)
        return _hip().packed_cumsum(feats.float(), starts, exclusive, reverse)

    @staticmethod
    def backward(ctx, grad):
        (starts,) = ctx.saved_tensors
        exclusive, reverse = ctx.flags
        return _hip().packed_cumsum(grad.contiguous().float(), starts, exclusive, not reverse), None, None, None


def cumsum(feats, boundary, exclusive=False, reverse=False, starts=None):
    """Running sum restarted at every pack boundary."""
    if starts is None:
        starts = pack_starts(boundary)
    return _Cumsum.apply(feats, starts, exclusive, reverse)


def exponential_integration(feats, tau, boundary, exclusive=True, starts=None):
    """(sum_reduce(w * feats), w) with w = exp(-cumsum(tau)) * (1 - exp(-tau))."""
    if starts is None:
        starts = pack_starts(boundary)
    alpha = 1.0 - torch.exp(-tau)
    w = torch.exp(-cumsum(tau, boundary, exclusive=exclusive, starts=starts)) * alpha
    return sum_reduce(w * feats, boundary, starts=starts), w


class _Composite(torch.autograd.Function):
    """Fused tau -> transmittance -> weights -> per-ray rgb / alpha / depth / hit (+ background)."""

    @staticmethod
    def forward(ctx, color, density, deltas, depths, ridx, starts, num_rays, bg):
        C = _hip()
        color32 = color.detach().float().contiguous()
        dens32 = density.detach().float().contiguous()
        rgb, alpha, depth, hit, _w = C.composite_fwd(color32, dens32, deltas, depths, ridx, starts, num_rays, bg)
        ctx.by_ray = ridx is None
        ctx.save_for_backward(color32, dens32, deltas, depths if depths is not None else torch.empty(0),
                              ridx if ridx is not None else torch.empty(0), starts)
        ctx.has_depth = depths is not None
        ctx.bg = bg
        ctx.in_dtypes = (color.dtype, density.dtype)
        ctx.mark_non_differentiable(hit)
        if depth is None:
            depth = torch.empty(0, device=rgb.device)
        return rgb, alpha, depth, hit

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth, _g_hit):
        color, density, deltas, depths, ridx, starts = ctx.saved_tensors
        depths = depths if ctx.has_depth else None
        ridx = None if ctx.by_ray else ridx
        C = _hip()
        if g_rgb is None:
            g_rgb = torch.zeros(g_alpha.shape[0], 3, device=color.device)
        gc, gd = C.composite_bwd(g_rgb.contiguous(), None if g_alpha is None else g_alpha.contiguous(),
                                 g_depth.contiguous() if (ctx.has_depth and g_depth is not None) else None,
                                 color, density, deltas, depths, ridx, starts, ctx.bg)
        return gc.to(ctx.in_dtypes[0]), gd.to(ctx.in_dtypes[1]), None, None, None, None, None, None


def composite(color, density, deltas, depths, ridx, starts, num_rays, bg):
    """-> (rgb [R,3], alpha [R,1], depth [R,1] or None, hit bool [R]); differentiable w.r.t. color and density.
    `starts` = pack starts [P] with `ridx` given, or per-ray sample offsets [R+1] with ridx=None (no compaction needed)."""
    rgb, alpha, depth, hit = _Composite.apply(color, density.reshape(-1, 1), deltas, depths, ridx, starts, num_rays, bg)
    return rgb, alpha, (depth if depths is not None else None), hit
