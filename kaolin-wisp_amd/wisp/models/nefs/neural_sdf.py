"""NeuralSDF: feature grid (+ optional position embedding) -> MLP -> signed distance (NGLOD).
Surface of wisp/models/nefs/neural_sdf.py:20-175."""
from typing import Any, Dict

import torch

from wisp.models.activations import get_activation_class
from wisp.models.decoders import BasicDecoder
from wisp.models.embedders import get_positional_embedder
from wisp.models.grids import BLASGrid
from wisp.models.layers import get_layer_class
from wisp.models.nefs.base_nef import BaseNeuralField


class NeuralSDF(BaseNeuralField):
    def __init__(self,
                 grid: BLASGrid,
                 pos_embedder: str = 'positional',  # options: 'none', 'identity', 'positional'
                 pos_multires: int = 4,
                 position_input: bool = True,
                 activation_type: str = 'relu',
                 layer_type: str = 'none',
                 hidden_dim: int = 128,
                 num_layers: int = 1
                 ):
        """
        Args:
            grid (BLASGrid): feature grid + occupancy structure (OctreeGrid for NGLOD).
            pos_embedder (str): 'none' | 'identity' | 'positional' embedding of the sample position.
            pos_multires (int): frequencies of the positional embedding.
            position_input (bool): feed the raw position to the decoder as well.
            activation_type (str), layer_type (str), hidden_dim (int), num_layers (int): decoder shape.
        """
        super().__init__()
        self.grid = grid
        self.pos_multires = pos_multires
        self.position_input = position_input
        self.pos_embedder, self.pos_embed_dim = self.init_embedder(pos_embedder, pos_multires, position_input)
        self.activation_type = activation_type
        self.layer_type = layer_type
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.decoder = self.init_decoder(activation_type, layer_type, num_layers, hidden_dim)

    def init_embedder(self, embedder_type, frequencies=None, position_input=True):
        if embedder_type == 'none' and not position_input:
            return None, 0
        if embedder_type == 'identity' or (embedder_type == 'none' and position_input):
            return torch.nn.Identity(), 3
        if embedder_type == 'positional':
            # the reference passes a keyword the embedder factory does not accept (neural_sdf.py:97); the intent
            # (include the input when position_input is set) is implemented here
            return get_positional_embedder(frequencies=frequencies, include_input=position_input)
        raise NotImplementedError(f'Unsupported embedder type for NeuralSDF: {embedder_type}')

    def init_decoder(self, activation_type, layer_type, num_layers, hidden_dim):
        return BasicDecoder(input_dim=self.decoder_input_dim(), output_dim=1, activation=get_activation_class(activation_type),
                            bias=True, layer=get_layer_class(layer_type), num_layers=num_layers, hidden_dim=hidden_dim, skip=[])

    def register_forward_functions(self):
        self._register_forward_function(self.sdf, ["sdf"])

    def sdf(self, coords, lod_idx=None):
        """coords [batch, num_samples, 3] or [batch, 3] -> dict(sdf [batch, (num_samples,) 1])."""
        shape = coords.shape
        if shape[0] == 0:
            return dict(sdf=torch.zeros_like(coords)[..., 0:1])
        if lod_idx is None:
            lod_idx = self.grid.num_lods - 1
        if len(shape) == 2:
            coords = coords[:, None]
        num_samples = coords.shape[1]
        feats = self.grid.interpolate(coords, lod_idx)
        if self.pos_embedder is not None:
            emb = self.pos_embedder(coords.reshape(-1, 3)).view(-1, num_samples, self.pos_embed_dim)
            feats = torch.cat([emb, feats], dim=-1)
        sdf = self.decoder(feats)
        if len(shape) == 2:
            sdf = sdf[:, 0]
        return dict(sdf=sdf)

    def effective_feature_dim(self):
        if self.grid.multiscale_type == 'cat':
            return self.grid.feature_dim * self.grid.num_lods
        return self.grid.feature_dim

    def decoder_input_dim(self):
        d = self.effective_feature_dim()
        if self.position_input:
            d += self.pos_embed_dim
        return d

    def public_properties(self) -> Dict[str, Any]:
        return {"Grid": self.grid, "Pos. Embedding": self.pos_embedder, "Decoder (sdf)": self.decoder}
