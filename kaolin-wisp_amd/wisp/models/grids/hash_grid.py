"""HashGrid: multi-resolution hashed feature grid (instant-ngp style) over an occupancy BLAS.
Constructor surface of wisp/models/grids/hash_grid.py:20-245; interpolation runs in csrc/hashgrid.hip."""
from __future__ import annotations

from typing import Any, Dict, List, Set, Type

import numpy as np
import torch

import wisp.ops.grid as grid_ops
import wisp.ops.spc as wisp_spc_ops
from wisp.accelstructs import BaseAS, OctreeAS, ASRaymarchResults
from wisp.models.grids.blas_grid import BLASGrid
from wisp.models.grids.utils import MultiTable


class HashGrid(BLASGrid):
    def __init__(self,
        blas               : BaseAS,
        feature_dim        : int,
        resolutions        : List[int],
        multiscale_type    : str    = 'sum',  # options: 'cat', 'sum'
        feature_std        : float  = 0.0,
        feature_bias       : float  = 0.0,
        codebook_bitwidth  : int    = 8,
        coord_dim          : int    = 3  # options: 2, 3
    ):
        """
        Args:
            blas (BaseAS): occupancy structure used for raymarching / queries (may be None for 2-D image fitting).
            feature_dim (int): features per table entry (even).
            resolutions (List[int]): grid resolution of every level of detail.
            multiscale_type (str): 'sum' adds the per-level features, 'cat' concatenates them.
            feature_std (float): std of the normal initialisation of the tables.
            feature_bias (float): stored for config compatibility (the reference never applies it).
            codebook_bitwidth (int): hashed levels have 2**codebook_bitwidth entries.
            coord_dim (int): 2 or 3.
        """
        super().__init__(blas)
        assert coord_dim in (2, 3)
        if self.blas is not None:
            self.dense_points = wisp_spc_ops.unbatched_get_level_points(
                self.blas.points, self.blas.pyramid, self.blas.max_level).clone()
            self.num_cells = self.dense_points.shape[0]
            self.occupancy = torch.zeros(self.num_cells)

        self.feature_dim = feature_dim
        self.multiscale_type = multiscale_type
        self.feature_std = feature_std
        self.feature_bias = feature_bias
        self.codebook_bitwidth = codebook_bitwidth

        self.resolutions = resolutions
        self.num_lods = len(resolutions)
        self.active_lods = [x for x in range(self.num_lods)]
        self.max_lod = self.num_lods - 1
        self.codebook_size = 2 ** self.codebook_bitwidth
        self.coord_dim = coord_dim
        self.codebook = MultiTable(resolutions, self.coord_dim, self.feature_dim, self.feature_std, self.codebook_size)

    @classmethod
    def from_octree(cls,
                    blas               : BaseAS,
                    feature_dim        : int,
                    base_lod           : int   = 2,
                    num_lods           : int   = 1,
                    multiscale_type    : str   = 'sum',   # options: 'cat', 'sum'
                    feature_std        : float = 0.0,
                    feature_bias       : float = 0.0,
                    codebook_bitwidth  : int   = 8,
                    coord_dim          : int   = 3) -> HashGrid:
        """Octree-style resolutions 2**base_lod, 2**(base_lod+1), ... (num_lods of them)."""
        resolutions = [2 ** (base_lod + x) for x in range(num_lods)]
        return cls(blas=blas, feature_dim=feature_dim, resolutions=resolutions, multiscale_type=multiscale_type,
                   feature_std=feature_std, feature_bias=feature_bias, codebook_bitwidth=codebook_bitwidth,
                   coord_dim=coord_dim)

    @classmethod
    def from_geometric(cls,
                       blas               : BaseAS,
                       feature_dim        : int,
                       num_lods           : int,
                       multiscale_type    : str = 'sum',    # options: 'cat', 'sum'
                       feature_std        : float = 0.0,
                       feature_bias       : float = 0.0,
                       codebook_bitwidth  : int   = 8,
                       min_grid_res       : int   = 16,
                       max_grid_res       : int   = None,
                       coord_dim          : int   = 3) -> HashGrid:
        """Geometric progression of resolutions between min_grid_res and max_grid_res (instant-ngp eq. 2-3)."""
        b = np.exp((np.log(max_grid_res) - np.log(min_grid_res)) / (num_lods - 1))
        resolutions = [int(np.floor(min_grid_res * (b ** l))) for l in range(num_lods)]
        return cls(blas=blas, feature_dim=feature_dim, resolutions=resolutions, multiscale_type=multiscale_type,
                   feature_std=feature_std, feature_bias=feature_bias, codebook_bitwidth=codebook_bitwidth,
                   coord_dim=coord_dim)

    @classmethod
    def from_resolutions(cls,
                         blas               : BaseAS,
                         feature_dim        : int,
                         resolutions        : List[int] = None,
                         multiscale_type    : str   = 'sum',  # options: 'cat', 'sum'
                         feature_std        : float = 0.0,
                         feature_bias       : float = 0.0,
                         codebook_bitwidth  : int   = 8,
                         coord_dim          : int   = 3) -> HashGrid:
        """Explicit list of per-level resolutions."""
        assert resolutions is not None, 'HashGrid.from_resolutions() constructor cannot accept a None resolutions arg.'
        return cls(blas=blas, feature_dim=feature_dim, resolutions=resolutions, multiscale_type=multiscale_type,
                   feature_std=feature_std, feature_bias=feature_bias, codebook_bitwidth=codebook_bitwidth,
                   coord_dim=coord_dim)

    def freeze(self):
        self.codebook.requires_grad_(False)

    def interpolate(self, coords, lod_idx):
        """coords [batch, num_samples, D] or [batch, D] -> features [..., feature_dim or num_lods*feature_dim].

        'cat' reproduces the reference's zeroing of columns lod_idx*feature_dim.. (hash_grid.py:226-229), fused
        into the kernel (zeroed columns are neither gathered nor given gradient)."""
        output_shape = coords.shape[:-1]
        if coords.ndim == 3:
            coords = coords.reshape(-1, coords.shape[-1])
        if self.multiscale_type == 'cat':
            feats = grid_ops.hashgrid(coords, self.codebook_bitwidth, lod_idx, self.codebook,
                                      zero_from_col=lod_idx * self.feature_dim)
            return feats.reshape(*output_shape, feats.shape[-1])
        elif self.multiscale_type == 'sum':
            feats = grid_ops.hashgrid(coords, self.codebook_bitwidth, lod_idx, self.codebook)
            return feats.reshape(*output_shape, self.num_lods, feats.shape[-1] // self.num_lods).sum(-2)
        raise NotImplementedError

    def raymarch(self, rays, raymarch_type, num_samples, level=None, **kwargs) -> ASRaymarchResults:
        """Samples are generated at the finest level of the occupancy structure (hash_grid.py:235-240)."""
        return self.blas.raymarch(rays, raymarch_type=raymarch_type, num_samples=num_samples,
                                  level=self.blas.max_level, **kwargs)

    def supported_blas(self) -> Set[Type[BaseAS]]:
        return {OctreeAS}

    def name(self) -> str:
        return "Hash Grid"

    def public_properties(self) -> Dict[str, Any]:
        parent = super().public_properties()
        return {**parent, "Feature Dims": self.feature_dim, "Total LODs": self.max_lod,
                "Active feature LODs": [str(x) for x in self.active_lods], "Interpolation": 'linear',
                "Multiscale aggregation": self.multiscale_type, "HashTable Size": f"2^{self.codebook_bitwidth}"}
