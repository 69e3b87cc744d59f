"""Geometric helpers (wisp/ops/geometric.py): the depth-bound search of the SDF tracer (:15-22) and the sphere samplers the
radiance field's prune draws its view directions from (:25-62)."""
import numpy as np
import torch


def find_depth_bound(query, nug_depth, info, curr_idxes=None):
    """For every ray, the nugget that contains (or first lies beyond) the query depth, searched forward from the
    ray's current nugget; -1 if none.  query [P,1], nug_depth [M,2], info = first-hit flags [M], curr_idxes int32 [P]."""
    import wisp._C as _C
    if curr_idxes is None:
        curr_idxes = torch.nonzero(info)[..., 0].int()
    return _C.find_depth_bound(query.reshape(-1).contiguous(), curr_idxes.contiguous(), nug_depth.contiguous())


def sample_unif_sphere(n):
    """n points uniformly on the unit sphere, float64 [n, 3] from numpy's global generator (wisp/ops/geometric.py:25-39):
    z uniform in [-1, 1], azimuth uniform in [0, 2 pi) - two draws per point, all z first."""
    u = np.random.rand(2, n)
    z = 1 - 2 * u[0, :]
    r = np.sqrt(1. - z * z)
    phi = 2 * np.pi * u[1, :]
    return np.array([r * np.cos(phi), r * np.sin(phi), z]).transpose()


def sample_fib_sphere(n):
    """n points spread evenly over the unit sphere by the golden-ratio spiral, in spiral order (wisp/ops/geometric.py:42-62)."""
    k = np.arange(0, n, dtype=float) + 0.5
    polar = np.arccos(1 - 2 * k / n)
    azimuth = 2. * np.pi * k / ((1 + 5 ** 0.5) / 2)
    return np.array([np.cos(azimuth) * np.sin(polar), np.sin(azimuth) * np.sin(polar), np.cos(polar)]).transpose()
