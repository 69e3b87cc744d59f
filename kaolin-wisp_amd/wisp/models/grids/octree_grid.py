"""OctreeGrid: multi-level feature grid on the corners of a sparse octree (NGLOD-style).
Constructor / attribute surface of wisp/models/grids/octree_grid.py:20-230; the per-level trilinear gather runs in
csrc/spc_interp.hip, the parent chain comes from the HIP octree query."""
from __future__ import annotations

from typing import Any, Dict, Set, Type

import torch
import torch.nn as nn

import wisp.ops.grid as grid_ops
import wisp.ops.spc as wisp_spc_ops
from wisp.accelstructs import BaseAS, OctreeAS, ASRaymarchResults
from wisp.models.grids.blas_grid import BLASGrid


class OctreeGrid(BLASGrid):
    def __init__(
        self,
        blas: BaseAS,
        feature_dim         : int,
        num_lods            : int          = 1,
        interpolation_type  : str = 'linear',   # options: 'linear', 'closest'
        multiscale_type     : str = 'cat',      # options: 'cat', 'sum'
        feature_std         : float        = 0.0,
        feature_bias        : float        = 0.0
    ):
        """
        Args:
            blas (BaseAS): occupancy octree; features live on the corners (dual octree) of its cells.
            feature_dim (int): features per corner.
            num_lods (int): number of feature levels, the finest being blas.max_level
                (base_lod = blas.max_level - num_lods + 1).
            interpolation_type (str): 'linear' (trilinear over the 8 corners) or 'closest' (one feature per cell).
            multiscale_type (str): 'cat' or 'sum' across levels.
            feature_std / feature_bias (float): N(feature_bias, feature_std) initialisation.
        """
        super().__init__(blas)
        self.feature_dim = feature_dim
        self.max_lod = blas.max_level
        self.num_lods = num_lods
        self.base_lod = self.max_lod - self.num_lods + 1
        self.interpolation_type = interpolation_type
        self.multiscale_type = multiscale_type
        self.feature_std = feature_std
        self.feature_bias = feature_bias
        self.active_lods = [self.base_lod + x for x in range(self.num_lods)]
        self.half_features = True     # reproduce the reference's fp16 rounding of features / results (octree_grid.py:147-149)
        if self.num_lods > 0:
            self.init_feature_structure()

    def _feature_pyramid(self):
        if self.interpolation_type == 'linear':
            self.points_dual, self.pyramid_dual, self.trinkets, self.parents = \
                wisp_spc_ops.make_trilinear_spc(self.blas.points, self.blas.pyramid)
            return [int(self.pyramid_dual[0, al]) + 1 for al in self.active_lods]
        if self.interpolation_type == 'closest':
            return [int(self.blas.pyramid[0, al]) + 1 for al in self.active_lods]
        raise Exception(f"Interpolation mode {self.interpolation_type} is not supported.")

    def init_feature_structure(self):
        """Dual octree + one [corners+1, feature_dim] parameter per active level."""
        sizes = self._feature_pyramid()
        self.num_feat = torch.tensor(sum(sizes)).long()
        self.features = nn.ParameterList([])
        for n in sizes:
            fts = torch.zeros(n, self.feature_dim) + self.feature_bias
            fts += torch.randn_like(fts) * self.feature_std
            self.features.append(nn.Parameter(fts))

    def freeze(self):
        for lod_idx in range(self.num_lods):
            self.features[lod_idx].requires_grad_(False)

    def _index_features(self, feats, idx):
        """Feature rows for corner indices (overridden by codebook grids)."""
        return feats[idx.long()]

    def _fusable(self):
        """The one-launch multi-level lookup covers plain trilinear feature tables (subclasses that re-index their
        features, e.g. the codebook grid, keep the per-level path)."""
        return (self.interpolation_type == 'linear' and type(self)._index_features is OctreeGrid._index_features
                and type(self)._interpolate is OctreeGrid._interpolate and self.multiscale_type in ('cat', 'sum')
                and len(self.active_lods) <= 16 and self.features[0].is_cuda)

    def _sync_device(self, device):
        if self.interpolation_type == 'linear' and self.trinkets.device != device:
            self.trinkets = self.trinkets.to(device)
        self.blas._to_device(device)

    def _interpolate(self, coords, feats, pidx, lod_idx):
        """coords [batch, num_samples, 3] inside voxels pidx [batch] of level active_lods[lod_idx] -> [batch, num_samples, C]."""
        batch, num_samples = coords.shape[:2]
        lod = self.active_lods[lod_idx]
        self._sync_device(coords.device)
        if self.interpolation_type == 'linear':
            return grid_ops.spc_interpolate_trilinear(coords, pidx, self.blas.points, self.trinkets.int(), feats, lod,
                                                      half_round=self.half_features)
        if self.interpolation_type == 'closest':
            fs = self._index_features(feats, pidx.long() - int(self.blas.pyramid[1, lod]))[..., None, :]
            return fs.expand(batch, num_samples, feats.shape[-1])
        raise Exception(f"Interpolation mode {self.interpolation_type} is not supported.")

    def interpolate(self, coords, lod_idx):
        """coords [batch, num_samples, 3] or [batch, 3] -> features at level index lod_idx ('cat' / 'sum' over 0..lod_idx)."""
        output_shape = coords.shape[:-1]
        if coords.ndim < 3:
            coords = coords[:, None]
        if lod_idx == 0:
            pidx = self.blas.query(coords[:, 0], self.active_lods[lod_idx], with_parents=False).pidx
            feat = self._interpolate(coords, self.features[0], pidx, 0)
            return feat.reshape(*output_shape, feat.shape[-1])
        num_feats = lod_idx + 1
        flat = coords.reshape(-1, 3)
        # the cell of every active level; an accelerator that only has the reference's surface (BaseAS.query) gives the same columns
        query_chain = getattr(self.blas, "query_chain", None)
        if query_chain is not None:
            chain = query_chain(flat, self.active_lods[lod_idx], self.base_lod)
        else:
            chain = self.blas.query(flat, self.active_lods[lod_idx], with_parents=True).pidx[..., self.base_lod:]
        if self._fusable():
            # all levels in one launch ('cat' row or the 'sum' written directly)
            self._sync_device(flat.device)
            feats = grid_ops.spc_interpolate_trilinear_multi(
                flat, chain, self.blas.points, self.trinkets.int(), [self.features[i] for i in range(num_feats)],
                self.active_lods[:num_feats], half_round=self.half_features, sum_lods=self.multiscale_type == 'sum')
            return feats.reshape(*output_shape, feats.shape[-1])
        feats = [self._interpolate(flat.reshape(-1, 1, 3), self.features[i], chain[:, i].contiguous(), i)[:, 0]
                 for i in range(num_feats)]
        feats = torch.cat(feats, dim=-1)
        if self.multiscale_type == 'sum':
            feats = feats.reshape(*feats.shape[:-1], num_feats, self.feature_dim).sum(-2)
            num_feats = 1
        return feats.reshape(*output_shape, self.feature_dim * num_feats)

    def raymarch(self, rays, raymarch_type, num_samples, level=None, **kwargs) -> ASRaymarchResults:
        """Samples are generated at the coarsest feature level (octree_grid.py:221-226)."""
        return self.blas.raymarch(rays, raymarch_type=raymarch_type, num_samples=num_samples, level=self.base_lod, **kwargs)

    def supported_blas(self) -> Set[Type[BaseAS]]:
        return {OctreeAS}

    def name(self) -> str:
        return "Octree Grid"

    def public_properties(self) -> Dict[str, Any]:
        parent = super().public_properties()
        active = None if not self.active_lods else f'{min(self.active_lods)} - {max(self.active_lods)}'
        return {**parent, "Feature Dims": self.feature_dim, "Total LODs": self.max_lod, "Active feature LODs": active,
                "Interpolation": self.interpolation_type, "Multiscale aggregation": self.multiscale_type}
