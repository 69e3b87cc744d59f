"""TriplanarGrid: features on a multi-resolution pyramid of axis-aligned plane triplets
(interface of wisp/models/grids/triplanar_grid.py:19-246: constructor schema, `features[i].fmx/fmy/fmz` parameter names and
shapes, 'cat' / 'sum' aggregation, AABB raymarch).  The reference evaluates three F.grid_sample calls per level and
stacks / concatenates / sums the results; here every level and plane is ONE HIP launch forward and one backward
(csrc/spc_interp.hip, `wisp_triplane_fwd/_bwd`) with grid_sample's align_corners=True + reflection semantics.
"""
from typing import Any, Dict, Set, Type

import torch
import torch.nn as nn

from wisp.accelstructs import BaseAS, AxisAlignedBBoxAS, ASRaymarchResults, ASRaytraceResults
from wisp.core import WispModule
from wisp.models.grids.blas_grid import BLASGrid


def _hip():
    import wisp._C as _C
    return _C


class _TriplaneLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, sum_lods, *planes):
        out = _hip().triplane_forward(coords.detach(), [p.detach() for p in planes], sum_lods)
        ctx.save_for_backward(coords.detach())
        ctx.meta = ([tuple(p.shape) for p in planes], [p.dtype for p in planes], sum_lods)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (coords,) = ctx.saved_tensors
        shapes, dtypes, sum_lods = ctx.meta
        grads = _hip().triplane_backward(coords, grad_out.contiguous().float(), shapes, sum_lods)
        return (None, None) + tuple(g.to(dt) for g, dt in zip(grads, dtypes))


def triplane_lookup(coords, volumes, sum_lods):
    """coords [N,3] -> [N, L*3*fdim] (or [N, 3*fdim] when summed) for a list of TriplanarFeatureVolume."""
    planes = [p for v in volumes for p in (v.fmx, v.fmy, v.fmz)]
    return _TriplaneLookup.apply(coords.contiguous(), sum_lods, *planes)


class TriplanarFeatureVolume(WispModule):
    """One level: three [1, fdim, fsize+1, fsize+1] feature maps (reference :185-203)."""

    def __init__(self, fdim, fsize, std, bias):
        super().__init__()
        self.fsize = fsize
        self.fdim = fdim
        self.fmx = nn.Parameter(torch.randn(1, fdim, fsize + 1, fsize + 1) * std + bias)
        self.fmy = nn.Parameter(torch.randn(1, fdim, fsize + 1, fsize + 1) * std + bias)
        self.fmz = nn.Parameter(torch.randn(1, fdim, fsize + 1, fsize + 1) * std + bias)
        self.padding_mode = 'reflection'

    def forward(self, x):
        """x [batch, num_samples, 3] -> [batch, 3, fdim, num_samples]; x [batch, 3] -> [batch, 3, fdim] (the reference's
        layouts, :214-233)."""
        flat = x.reshape(-1, 3)
        f = triplane_lookup(flat, [self], False)                         # [N, 3*fdim]
        if x.ndim == 3:
            return f.reshape(x.shape[0], x.shape[1], 3, self.fdim).permute(0, 2, 3, 1)
        return f.reshape(x.shape[0], 3, self.fdim)

    def name(self) -> str:
        return "TriplanarFeatureVolume"

    def public_properties(self) -> Dict[str, Any]:
        return {'Resolution': f'3x{self.fsize}x{self.fsize}'}


class TriplanarGrid(BLASGrid):
    def __init__(self,
                 blas: BaseAS,
                 feature_dim: int,
                 log_base_resolution: int = 4,
                 num_lods: int = 1,
                 interpolation_type: str = 'linear',  # options: 'linear', 'closest'
                 multiscale_type: str = 'sum',  # options: 'cat', 'sum'
                 feature_std: float = 0.0,
                 feature_bias: float = 0.0
                 ):
        """blas: the AABB the planes span; feature_dim: features PER PLANE (the grid's feature_dim is 3x that, as in the
        reference :66); level i has planes of (2^(log_base_resolution + i) + 1)^2 texels."""
        super().__init__(blas=blas)
        self.feature_dim = feature_dim * 3
        self.num_lods = num_lods
        self.log_base_resolution = log_base_resolution
        self.interpolation_type = interpolation_type
        self.multiscale_type = multiscale_type
        self.feature_std = feature_std
        self.feature_bias = feature_bias
        self.active_lods = [log_base_resolution + x for x in range(self.num_lods)]
        self.num_feat = 0
        self.init_feature_structure()

    def init_feature_structure(self):
        self.features = nn.ModuleList([])
        self.num_feat = 0
        for i in self.active_lods:
            self.features.append(TriplanarFeatureVolume(self.feature_dim // 3, 2 ** i, self.feature_std, self.feature_bias))
            self.num_feat += ((2 ** i + 1) ** 2) * self.feature_dim * 3

    def freeze(self):
        self.features.requires_grad_(False)

    def interpolate(self, coords, lod_idx):
        """coords [batch, num_samples, 3] or [batch, 3] -> [..., feature_dim] ('sum') or [..., (lod_idx+1) * feature_dim] ('cat').
        Like the reference (triplanar_grid.py:110-122), which inflates [batch, 3] to [batch, 1, 3] and restores the caller's shape in
        its 'sum' branch only, 'cat' answers a [batch, 3] query with [batch, 1, width]; NeuralRadianceField.rgba reshapes either."""
        if self.interpolation_type != 'linear':
            raise ValueError(f"Interpolation mode '{self.interpolation_type}' is not supported")
        output_shape = coords.shape[:-1]
        feats = triplane_lookup(coords.reshape(-1, 3), [self.features[i] for i in range(lod_idx + 1)],
                                self.multiscale_type == 'sum')
        if self.multiscale_type != 'sum' and coords.ndim < 3:
            return feats.reshape(coords.shape[0], 1, feats.shape[-1])
        return feats.reshape(*output_shape, feats.shape[-1])

    def _interpolate(self, coords, feats, lod_idx):
        """coords [batch, num_samples, 3], feats = one TriplanarFeatureVolume -> [batch, num_samples, 3 * fdim]."""
        if self.interpolation_type != 'linear':
            raise ValueError(f"Interpolation mode '{self.interpolation_type}' is not supported")
        batch, num_samples = coords.shape[:2]
        return triplane_lookup(coords.reshape(-1, 3), [feats], False).reshape(batch, num_samples, 3 * feats.fdim)

    def raymarch(self, rays, raymarch_type, num_samples, level=None, **kwargs) -> ASRaymarchResults:
        """The BLAS is only an AABB tracer here (reference :148-153)."""
        return self.blas.raymarch(rays, raymarch_type=raymarch_type, num_samples=num_samples, level=0, **kwargs)

    def raytrace(self, rays, level=None, with_exit=False) -> ASRaytraceResults:
        return self.blas.raytrace(rays, level=0, with_exit=with_exit)

    def supported_blas(self) -> Set[Type[BaseAS]]:
        return {AxisAlignedBBoxAS}

    def name(self) -> str:
        return "Triplanar Grid"

    def public_properties(self) -> Dict[str, Any]:
        parent = super().public_properties()
        active = None if not self.active_lods else f'{min(self.active_lods)} - {max(self.active_lods)}'
        return {**parent, "Feature Dims": self.feature_dim, "Total LODs": self.num_lods, "Active feature LODs": active,
                "Interpolation": self.interpolation_type, "Multiscale aggregation": self.multiscale_type}
