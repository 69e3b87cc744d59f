from .ray_sampler import SampleRays
