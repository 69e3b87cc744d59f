"""Geometric helpers of the SDF path (wisp/ops/geometric.py:15-22)."""
import torch


def find_depth_bound(query, nug_depth, info, curr_idxes=None):
    """For every ray, the nugget that contains (or first lies beyond) the query depth, searched forward from the
    ray's current nugget; -1 if none.  query [P,1], nug_depth [M,2], info = first-hit flags [M], curr_idxes int32 [P]."""
    import wisp._C as _C
    if curr_idxes is None:
        curr_idxes = torch.nonzero(info)[..., 0].int()
    return _C.find_depth_bound(query.reshape(-1).contiguous(), curr_idxes.contiguous(), nug_depth.contiguous())
