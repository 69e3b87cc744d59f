"""SampleRays: sub-sample a fixed number of rays of a view, on the device the rays live on
(wisp/datasets/transforms/ray_sampler.py:13-35).  The trainer re-sizes it every step (calc_adaptive_rays)."""
import torch

from wisp.core import Rays
from wisp.datasets.batch import MultiviewBatch


class SampleRays:
    def __init__(self, num_samples: int):
        self.num_samples = num_samples

    def set_num_samples(self, num_samples: int):
        self.num_samples = num_samples

    def __call__(self, inputs: MultiviewBatch, generator=None):
        # the reference's profiler range (ray_sampler.py:24: @torch.cuda.nvtx.range("SampleRays"); roctx on ROCm), as a plain push /
        # pop pair: the decorator form goes through contextlib on every call - 8-17 us of host time in a loop that is host bound
        torch.cuda.nvtx.range_push("SampleRays")
        try:
            return self._sample(inputs, generator)
        finally:
            torch.cuda.nvtx.range_pop()

    def _sample(self, inputs: MultiviewBatch, generator=None):
        rays = inputs['rays']
        ray_idx = torch.randint(0, rays.shape[0], [self.num_samples], device=rays.origins.device, generator=generator)
        values = inputs.ray_values() if hasattr(inputs, 'ray_values') else {k: v for k, v in inputs.items() if k != 'rays'}
        names = list(values)
        tensors = [rays.origins, rays.dirs] + [values[k] for k in names]
        scalar_bounds = not (torch.is_tensor(rays.dist_min) or torch.is_tensor(rays.dist_max))
        fused = (rays.origins.is_cuda and scalar_bounds and rays.origins.dim() == 2 and len(tensors) <= 4
                 and all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                         and t.shape[0] == rays.origins.shape[0] for t in tensors))
        if fused:                                  # one HIP launch gathers every per-ray tensor (wisp_gather_rows)
            import wisp._C as _C
            got = _C.gather_rows(ray_idx, tensors)
            out = {'rays': Rays(got[0], got[1], dist_min=rays.dist_min, dist_max=rays.dist_max)}
            for name, g in zip(names, got[2:]):
                out[name] = g
            return out
        out = {'rays': rays[ray_idx].contiguous()}
        for name, value in values.items():
            out[name] = value[ray_idx].contiguous()
        return out
