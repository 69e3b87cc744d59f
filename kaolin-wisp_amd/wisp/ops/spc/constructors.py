"""SPC constructors: dense octree and the dual ("trilinear") octree used by OctreeGrid.
Counterparts of wisp/ops/spc/constructors.py:14-47 (which defer to Kaolin-Core's make_dual / make_trinkets)."""
import torch

from .conversions import default_device, morton_to_points, points_to_morton


def create_dense_octree(level):
    """All 8^level cells occupied: every non-leaf node byte is 0xFF (sum_{l<level} 8^l bytes)."""
    n = sum(8 ** l for l in range(level))
    return torch.full((n,), 255, dtype=torch.uint8, device=default_device())


_OFFS = [[(j >> 2) & 1, (j >> 1) & 1, j & 1] for j in range(8)]


def make_trilinear_spc(points, pyramid):
    """-> (points_dual int16, pyramid_dual int32 [2,L+2], trinkets int32 [P,8], parents int32 [P]).
    Dual level l = unique corners p + {0,1}^3 of the level-l voxels, morton sorted; trinkets[p, j] indexes
    corner j = dx<<2|dy<<1|dz inside level l's dual block."""
    level = pyramid.shape[1] - 2
    dev = points.device
    offs = torch.tensor(_OFFS, dtype=torch.int64, device=dev)
    duals, counts, trinkets, parents = [], [], [], []
    for l in range(level + 1):
        s, n = int(pyramid[1, l]), int(pyramid[0, l])
        p = points[s:s + n].long()
        cm = points_to_morton((p[:, None, :] + offs[None]).reshape(-1, 3))
        dm = torch.unique(cm)
        duals.append(morton_to_points(dm))
        counts.append(dm.shape[0])
        trinkets.append(torch.searchsorted(dm, cm).reshape(n, 8).int())
        if l == 0:
            parents.append(torch.full((n,), -1, dtype=torch.int32, device=dev))
        else:
            ps, pn = int(pyramid[1, l - 1]), int(pyramid[0, l - 1])
            pm = points_to_morton(points[ps:ps + pn])
            parents.append((ps + torch.searchsorted(pm, points_to_morton(p) >> 3)).int())
    pyramid_dual = torch.zeros(2, level + 2, dtype=torch.int32)
    pyramid_dual[0, :level + 1] = torch.tensor(counts, dtype=torch.int32)
    pyramid_dual[1, 1:] = torch.cumsum(pyramid_dual[0, :-1], 0)
    return torch.cat(duals), pyramid_dual, torch.cat(trinkets), torch.cat(parents)
