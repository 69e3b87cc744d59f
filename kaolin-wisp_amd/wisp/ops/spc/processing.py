"""Point-set processing helpers (wisp/ops/spc/processing.py:13-47)."""
import itertools

import torch

from .conversions import morton_to_points, points_to_morton


def dilate_points(points, level):
    """Dilation of quantised points of `level` exactly as the reference spells it out (wisp/ops/spc/processing.py:26-41): 6 face,
    8 corner and 9 of the 12 edge neighbours - its list has no -x-y, -x-z, -y-z term and no term for the points themselves - so a
    cell grows into 23 neighbours.  Clipped to the grid, unique, morton sorted."""
    skipped = ((0, 0, 0), (-1, -1, 0), (-1, 0, -1), (0, -1, -1))
    shifts = [s for s in itertools.product((-1, 0, 1), repeat=3) if s not in skipped]
    offs = torch.tensor(shifts, dtype=torch.int16, device=points.device)
    grown = (points[None, :, :] + offs[:, None, :]).reshape(-1, 3)
    grown = torch.clip(grown, 0, 2 ** level - 1)
    return morton_to_points(torch.unique(points_to_morton(grown)))
