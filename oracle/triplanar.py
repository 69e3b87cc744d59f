"""TEST INFRASTRUCTURE ONLY - TriplanarGrid on the CPU exactly as the reference evaluates it: three
torch.nn.functional.grid_sample(align_corners=True, padding_mode='reflection') calls per level
(wisp/models/grids/triplanar_grid.py:205-233), stacked [x | y | z], then cat / sum over levels (:97-124).  torch's CPU
grid_sample IS the reference's arithmetic for this op, so parity here is pinned to it.  Also PINNED to TriplanarFeatureVolume.forward and
TriplanarGrid.interpolate / _interpolate compiled from the reference file (plane addressing and layout; 1e-6)."""
import torch
import torch.nn.functional as F


def volume_forward(fmx, fmy, fmz, x):
    """x [N,3] -> [N, 3, fdim] (the reference's 2-D branch has a typo'd keyword at :229; this is what it means)."""
    N = x.shape[0]
    g = x.reshape(1, N, 1, 3)
    sx = F.grid_sample(fmx, g[..., [1, 2]], align_corners=True, padding_mode='reflection')[0, :, :, 0].transpose(0, 1)
    sy = F.grid_sample(fmy, g[..., [0, 2]], align_corners=True, padding_mode='reflection')[0, :, :, 0].transpose(0, 1)
    sz = F.grid_sample(fmz, g[..., [0, 1]], align_corners=True, padding_mode='reflection')[0, :, :, 0].transpose(0, 1)
    return torch.stack([sx, sy, sz], dim=1)


def interpolate(volumes, coords, lod_idx, multiscale_type):
    """volumes: list of (fmx, fmy, fmz); coords [N,3] -> [N, F]."""
    N = coords.shape[0]
    feats = [volume_forward(*volumes[i], coords).reshape(N, -1) for i in range(lod_idx + 1)]
    feats = torch.cat(feats, dim=-1)
    if multiscale_type == 'sum':
        feats = feats.reshape(N, lod_idx + 1, feats.shape[-1] // (lod_idx + 1)).sum(-2)
    return feats
