"""Depth-interval sampling helpers of the reference's public ops API (wisp/ops/spc/sampling.py:35-71; SURVEY.md 2.1 lists both
as API).  NOTHING in this package calls them: the 'voxel' march computes the same quantities inside csrc/raymarch.hip
(wisp_raymarch_voxel_emit), bit for bit (tests/test_gpu_0_parity.py::test_raymarch_voxel_and_uniform_bit_exact).  They exist so
that user code importing `wisp.ops.spc.sample_from_depth_intervals / expand_pack_boundary` keeps working, and the CPU suite holds
them to `torch.equal` with the reference's functions (same random draw, same float operations in the same order) - which pins
the arithmetic: (k + u) * (1 / N), then entry + (exit - entry) * that, in fp32, one rounding per operation."""
import torch


def sample_from_depth_intervals(depth_intervals, num_samples, jitter=None):
    """[M, 2] (entry, exit) depths of M nuggets -> [M, num_samples] depths, one per stratum: entry + (exit - entry) * (k + u) / N
    with u ~ U[0, 1) drawn as ONE [M, N] fp32 tensor from the default generator (the reference's draw), or `jitter` when given."""
    entry, exit_ = depth_intervals[..., 0:1], depth_intervals[..., 1:2]
    shape = (depth_intervals.shape[0], num_samples)
    u = torch.rand(shape, dtype=torch.float32, device=depth_intervals.device) if jitter is None else jitter
    strata = torch.arange(num_samples, dtype=torch.float32, device=depth_intervals.device).expand(shape)
    frac = (strata + u) * (1.0 / num_samples)
    return entry + (exit_ - entry) * frac


def expand_pack_boundary(pack_boundary, num_samples):
    """boundary flags of M nuggets -> int32 flags of M * num_samples samples: the FIRST sample of a flagged nugget starts a pack."""
    first = pack_boundary.bool().reshape(-1, 1)
    rest = first.new_zeros(first.shape[0], num_samples - 1)
    return torch.cat([first, rest], dim=1).reshape(-1).int()
