"""CodebookOctreeGrid (VQAD): octree corners store logits over a small dictionary of feature vectors.
Surface of wisp/models/grids/codebook_grid.py:21-200.  Selection (straight-through softmax one-hot / argmax) and the
trilinear blend run as ONE fused HIP kernel per level (csrc/spc_interp.hip) when the dictionary is small enough
(<= 256 entries, <= 16 features, fp32); otherwise the reference formulation on gathered logits is used."""
from typing import Any, Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

import wisp.ops.grid as grid_ops
from wisp.accelstructs import BaseAS
from wisp.models.grids.octree_grid import OctreeGrid


class CodebookOctreeGrid(OctreeGrid):
    def __init__(
        self,
        blas                : BaseAS,
        feature_dim         : int,
        num_lods            : int          = 1,
        interpolation_type  : str          = 'linear',  # options: 'linear', 'closest'
        multiscale_type     : str          = 'cat',
        feature_std         : float        = 0.0,
        feature_bias        : float        = 0.0,
        codebook_bitwidth   : int          = 8
    ):
        """Same arguments as OctreeGrid plus codebook_bitwidth: every level has a dictionary of 2**bitwidth vectors."""
        self.bitwidth = codebook_bitwidth
        self.fused = True             # fused selection + blend kernel (csrc/spc_interp.hip)
        super().__init__(blas=blas, feature_dim=feature_dim, num_lods=num_lods, interpolation_type=interpolation_type,
                         multiscale_type=multiscale_type, feature_std=feature_std, feature_bias=feature_bias)

    def init_feature_structure(self):
        sizes = self._feature_pyramid()
        self.num_feat = torch.tensor(sum(sizes)).long()
        self.dictionary_size = 2 ** self.bitwidth
        self.dictionary = nn.ParameterList([
            nn.Parameter(torch.randn(self.dictionary_size, self.feature_dim) * self.feature_std) for _ in self.active_lods])
        self.features = nn.ParameterList([
            nn.Parameter(torch.randn(n, self.dictionary_size) * self.feature_std) for n in sizes])

    def bake(self):
        """Replace the logits by the selected dictionary index (inference-only compression)."""
        for i, f in enumerate(self.features):
            self.features[i] = nn.Parameter(f.max(dim=-1)[1].float())

    def _index_features(self, feats, idx, lod_idx):
        """Dictionary vectors for corner indices: straight-through one-hot of softmax(logits) in training,
        plain argmax lookup in eval (codebook_grid.py:103-136)."""
        if self.training:
            logits = feats[idx.long()]
            y_soft = F.softmax(logits, dim=-1)
            index = y_soft.max(-1, keepdim=True)[1]
            y_hard = torch.zeros_like(logits).scatter_(-1, index, 1.0)
            keys = y_hard - y_soft.detach() + y_soft
            return (self.dictionary[lod_idx][None, None] * keys[..., None]).sum(-2)
        keys = torch.max(feats[idx.long()], dim=-1)[1]
        return self.dictionary[lod_idx][keys]

    def _interpolate(self, coords, feats, pidx, lod_idx):
        batch, num_samples = coords.shape[:2]
        self._sync_device(coords.device)
        if self.interpolation_type != 'linear':
            if self.interpolation_type == 'closest':
                raise NotImplementedError
            raise Exception(f"Interpolation mode {self.interpolation_type} is not supported.")
        dictionary = self.dictionary[lod_idx]
        # (under autocast too: the fused op computes in fp32 whatever the ambient autocast dtype - the softmax the
        # reference runs here is an fp32-autocast op as well, codebook_grid.py:117-125)
        if (self.fused and coords.is_cuda and feats.dtype == torch.float32 and feats.ndim == 2
                and dictionary.shape[1] <= dictionary.shape[0] <= 256 and dictionary.shape[1] <= 16):
            return grid_ops.codebook_interpolate_trilinear(coords, pidx, self.blas.points, self.trinkets.int(), feats,
                                                           dictionary, self.active_lods[lod_idx], self.training)
        fs = torch.zeros(batch, num_samples, self.feature_dim, device=coords.device)
        valid = pidx > -1
        vp = pidx[valid].long()
        if vp.shape[0] == 0:
            return fs
        corner_feats = self._index_features(feats, self.trinkets.index_select(0, vp).long(), lod_idx)[:, None]   # [V,1,8,F]
        pts = self.blas.points.index_select(0, vp)
        coeffs = grid_ops.coords_to_trilinear_coeffs(coords[valid], pts, self.active_lods[lod_idx])[..., None]    # [V,S,8,1]
        fs[valid] = (corner_feats * coeffs).sum(-2)
        return fs

    def name(self) -> str:
        return "Codebook Grid"

    def public_properties(self) -> Dict[str, Any]:
        return {**super().public_properties(), "Bitwidth": self.bitwidth}
