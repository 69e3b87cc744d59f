"""Depth-interval sampling helpers kept for API compatibility (wisp/ops/spc/sampling.py:35-71).
The tracer's hot path does not call these - 'voxel' raymarch is fused in csrc/raymarch.hip."""
import torch


def sample_from_depth_intervals(depth_intervals, num_samples, jitter=None):
    """[M,2] (entry, exit) -> [M, num_samples] jittered depths: entry + (exit-entry) * (k + u) / N."""
    k = torch.arange(num_samples, device=depth_intervals.device)[None].float().repeat([depth_intervals.shape[0], 1])
    k += torch.rand_like(k) if jitter is None else jitter
    k *= (1.0 / num_samples)
    return depth_intervals[..., 0:1] + (depth_intervals[..., 1:2] - depth_intervals[..., 0:1]) * k


def expand_pack_boundary(pack_boundary, num_samples):
    """boundary flags of M nuggets -> flags of M*num_samples samples (first sample of each flagged nugget)."""
    out = torch.zeros(pack_boundary.shape[0] * num_samples, device=pack_boundary.device).bool()
    out[pack_boundary.nonzero().long() * num_samples] = True
    return out.int()
