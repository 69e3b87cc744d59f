"""Point-set processing helpers (wisp/ops/spc/processing.py:13-47)."""
import itertools

import torch

from .conversions import morton_to_points, points_to_morton


def dilate_points(points, level):
    """26-neighbourhood dilation of quantised points of `level`, clipped to the grid, morton sorted."""
    shifts = [s for s in itertools.product((-1, 0, 1), repeat=3) if s != (0, 0, 0)]
    offs = torch.tensor(shifts, dtype=torch.int16, device=points.device)
    grown = (points[None, :, :] + offs[:, None, :]).reshape(-1, 3)
    grown = torch.clip(grown, 0, 2 ** level - 1)
    return morton_to_points(torch.unique(points_to_morton(grown)))
