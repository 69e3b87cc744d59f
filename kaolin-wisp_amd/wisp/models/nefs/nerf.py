"""NeuralRadianceField: feature grid -> density MLP -> (geometry features + embedded view dir) -> colour MLP.

Constructor / method surface of wisp/models/nefs/nerf.py:20-300.  On MI355X the decoder pair runs as ONE fused
HIP kernel (csrc/nerf_mlp.hip: both MLPs, the view-direction positional encoding, relu / sigmoid epilogues)
whenever the configuration is the one the shipped configs use (relu, nn.Linear, one hidden density layer, two
hidden colour layers, 'positional' view embedding, no position embedding); anything else takes the generic
torch-module path (rocBLAS GEMMs).
"""
from typing import Any, Dict, Optional

import numpy as np
import math

import torch

from wisp.models.activations import get_activation_class
from wisp.models.decoders import BasicDecoder
from wisp.models.embedders import get_positional_embedder
from wisp.models.grids import BLASGrid, HashGrid
from wisp.models.layers import get_layer_class
from wisp.models.nefs.base_nef import BaseNeuralField
from wisp.ops.geometric import sample_unif_sphere


class NeuralRadianceField(BaseNeuralField):
    def __init__(self,
                 grid: BLASGrid,
                 # embedder args
                 pos_embedder: str = 'none',    # options: 'none', 'identity', 'positional'
                 view_embedder: str = 'none',   # options: 'none', 'identity', 'positional'
                 pos_multires: int = 10,
                 view_multires: int = 4,
                 position_input: bool = False,
                 # decoder args
                 activation_type: str = 'relu', #  options: 'none', 'relu', 'sin', 'fullsort', 'minmax'
                 layer_type: str = 'linear',    # 'linear', 'spectral_norm', 'frobenius_norm', 'l_1_norm', 'l_inf_norm'
                 hidden_dim: int = 128,
                 num_layers: int = 1,
                 bias: bool = False,
                 # pruning args
                 prune_density_decay: Optional[float] = (0.01 * 512) / np.sqrt(3),
                 prune_min_density: Optional[float] = 0.6,
                 ):
        """
        Args:
            grid (BLASGrid): feature grid + occupancy structure.
            pos_embedder / view_embedder (str): 'none' | 'identity' | 'positional' embedding of the sample position /
                view direction.
            pos_multires / view_multires (int): number of frequencies of the positional embeddings.
            position_input (bool): also feed the raw position to the density decoder.
            activation_type (str), layer_type (str), hidden_dim (int), num_layers (int), bias (bool): decoder shape.
            prune_density_decay (float), prune_min_density (float): occupancy pruning parameters (instant-ngp scheme).
        """
        super().__init__()
        self.grid = grid
        self.pos_embedder_type = pos_embedder
        self.view_embedder_type = view_embedder
        self.pos_multires = pos_multires
        self.view_multires = view_multires
        self.pos_embedder, self.pos_embed_dim = self.init_embedder(pos_embedder, pos_multires,
                                                                   include_input=position_input)
        self.view_embedder, self.view_embed_dim = self.init_embedder(view_embedder, view_multires, include_input=True)
        self.activation_type = activation_type
        self.layer_type = layer_type
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.bias = bias
        self.decoder_density, self.decoder_color = self.init_decoders(activation_type, layer_type, num_layers, hidden_dim)
        self.prune_density_decay = prune_density_decay
        self.prune_min_density = prune_min_density
        # MI355X backend knobs (not part of the reference schema)
        self.fused_decoder = True            # use csrc/nerf_mlp.hip when the configuration allows it
        self.decoder_compute = 'auto'        # 'fp32' | 'bf16' | 'auto' (bf16 under autocast, else fp32)

    def init_embedder(self, embedder_type, frequencies=None, include_input=False):
        if embedder_type == 'none' and not include_input:
            return None, 0
        if embedder_type == 'identity' or (embedder_type == 'none' and include_input):
            return torch.nn.Identity(), 3
        if embedder_type == 'positional':
            return get_positional_embedder(frequencies=frequencies, include_input=include_input)
        raise NotImplementedError(f'Unsupported embedder type for NeuralRadianceField: {embedder_type}')

    def init_decoders(self, activation_type, layer_type, num_layers, hidden_dim):
        act, layer = get_activation_class(activation_type), get_layer_class(layer_type)
        density = BasicDecoder(input_dim=self.density_net_input_dim(), output_dim=16, activation=act, bias=self.bias,
                               layer=layer, num_layers=num_layers, hidden_dim=hidden_dim, skip=[])
        if density.lout.bias is not None:
            density.lout.bias.data[0] = 1.0          # start with non-zero density (nerf.py:162-163)
        color = BasicDecoder(input_dim=self.color_net_input_dim(), output_dim=3, activation=act, bias=self.bias,
                             layer=layer, num_layers=num_layers + 1, hidden_dim=hidden_dim, skip=[])
        return density, color

    # ------------------------------------------------------------------ pruning
    def prune(self, unit_samples=None, view_dirs=None):
        """Occupancy update + BLAS rebuild (nerf.py:175-212).  `unit_samples` ([cells,3] in [0,1)) and `view_dirs`
        can be injected for reproducibility; by default they are drawn like the reference does."""
        if self.prune_density_decay is None or self.prune_min_density is None or self.grid is None:
            return
        if not isinstance(self.grid, HashGrid):
            raise NotImplementedError(f'Pruning not implemented for grid type {self.grid.__class__.__name__}')
        device = self.device
        self.grid.occupancy = self.grid.occupancy.to(device) * self.prune_density_decay
        if self.grid.dense_points.device != device:
            self.grid.dense_points = self.grid.dense_points.to(device)       # once: 12 MB at level 7, not per prune
        points = self.grid.dense_points
        res = 2.0 ** self.grid.blas.max_level
        if unit_samples is None:
            unit_samples = torch.rand(points.shape[0], 3, device=device)
        samples = ((points.float() + unit_samples.to(device)) / res) * 2.0 - 1.0
        if view_dirs is None:
            # uniform on the sphere like sample_unif_sphere (z uniform, azimuth uniform), but drawn on the device: the reference
            # builds 2.1 M directions with numpy on the host and copies them over (nerf.py:196; 90 ms per prune here, ~1 ms per
            # step of an unchanged trainer) for a query whose only consumed channel, the density, does not depend on them
            u = torch.rand(2, samples.shape[0], device=device)
            z = 1.0 - 2.0 * u[0]
            r = torch.sqrt((1.0 - z * z).clamp_min(0.0))
            phi = (2.0 * math.pi) * u[1]
            view_dirs = torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], dim=-1)
        with torch.no_grad():
            density = self.forward(coords=samples, ray_d=view_dirs.to(device), channels="density")
        self.grid.occupancy = torch.maximum(density[:, 0].float(), self.grid.occupancy)
        keep = self.grid.occupancy > self.prune_min_density
        blas_cls = self.grid.blas.__class__
        if not hasattr(blas_cls, "from_quantized_points"):
            raise Exception(f"The BLAS {blas_cls.__name__} does not support initialization "
                            "from_quantized_points, which is required for pruning.")
        level = self.grid.blas.max_level
        if (hasattr(blas_cls, "from_leaf_mask") and keep.is_cuda and points.shape[0] == 8 ** level and level <= 9):
            # dense_points are ALL cells of the level in hierarchy (= Morton) order, so `keep` already is the leaf mask of
            # the new octree: no boolean gather, no sort, no size read-back before the build
            new_blas = blas_cls.from_leaf_mask(keep, level)
            if new_blas is not None:
                self.grid.blas = new_blas
            return
        kept = points[keep]
        if kept.shape[0] == 0:
            return
        self.grid.blas = blas_cls.from_quantized_points(kept, level)

    # ------------------------------------------------------------------ forward
    def register_forward_functions(self):
        self._register_forward_function(self.rgba, ["density", "rgb"])

    def _can_fuse(self, feats):
        if not self.fused_decoder:
            return False
        from wisp.ops.nerf_mlp import supports
        return supports(self, feats)

    def rgba(self, coords, ray_d, lod_idx=None):
        """coords [batch,3], ray_d [batch,3] -> dict(rgb [batch,3] in [0,1], density [batch,1])."""
        if lod_idx is None:
            lod_idx = len(self.grid.active_lods) - 1
        batch, _ = coords.shape
        feats = self.grid.interpolate(coords, lod_idx).reshape(batch, self.effective_feature_dim())

        if self._can_fuse(feats):
            from wisp.ops.nerf_mlp import fused_nerf_decoder
            rgb, density = fused_nerf_decoder(self, feats, ray_d)
            return dict(rgb=rgb, density=density)

        if self.pos_embedder is not None:
            feats = torch.cat([feats, self.pos_embedder(coords).view(batch, self.pos_embed_dim)], dim=-1)
        density_feats = self.decoder_density(feats)
        if self.view_embedder is not None:
            fdir = torch.cat([density_feats, self.view_embedder(ray_d).view(batch, self.view_embed_dim)], dim=-1)
        else:
            fdir = density_feats
        colors = torch.sigmoid(self.decoder_color(fdir[..., 1:]))      # density_feats[0] is the density logit
        density = torch.relu(density_feats[..., 0:1])
        return dict(rgb=colors, density=density)

    def effective_feature_dim(self):
        if self.grid.multiscale_type == 'cat':
            return self.grid.feature_dim * self.grid.num_lods
        return self.grid.feature_dim

    def density_net_input_dim(self):
        return self.effective_feature_dim() + self.pos_embed_dim

    def color_net_input_dim(self):
        return 15 + self.view_embed_dim

    def public_properties(self) -> Dict[str, Any]:
        props = {"Grid": self.grid, "Pos. Embedding": self.pos_embedder, "View Embedding": self.view_embedder,
                 "Decoder (density)": self.decoder_density, "Decoder (color)": self.decoder_color}
        if self.prune_density_decay is not None:
            props['Pruning Density Decay'] = self.prune_density_decay
        if self.prune_min_density is not None:
            props['Pruning Min Density'] = self.prune_min_density
        return props
