"""SampleRays: sub-sample a fixed number of rays of a view, on the device the rays live on
(wisp/datasets/transforms/ray_sampler.py:13-35).  The trainer re-sizes it every step (calc_adaptive_rays)."""
import torch

from wisp.datasets.batch import MultiviewBatch


class SampleRays:
    def __init__(self, num_samples: int):
        self.num_samples = num_samples

    def set_num_samples(self, num_samples: int):
        self.num_samples = num_samples

    def __call__(self, inputs: MultiviewBatch, generator=None):
        rays = inputs['rays']
        ray_idx = torch.randint(0, rays.shape[0], [self.num_samples], device=rays.origins.device, generator=generator)
        out = {'rays': rays[ray_idx].contiguous()}
        values = inputs.ray_values() if hasattr(inputs, 'ray_values') else {k: v for k, v in inputs.items() if k != 'rays'}
        for name, value in values.items():
            out[name] = value[ray_idx].contiguous()
        return out
