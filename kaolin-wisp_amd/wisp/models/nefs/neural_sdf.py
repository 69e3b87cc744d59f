"""NeuralSDF (NGLOD): octree feature grid (+ optional embedded position) -> MLP -> signed distance.
Mirrors the surface of wisp/models/nefs/neural_sdf.py:20-175 (constructor schema, `sdf` channel, introspection); the
construction itself lives in _grid_mlp.py."""
from typing import Any, Dict

from wisp.models.grids import BLASGrid
from wisp.models.nefs import _grid_mlp
from wisp.models.nefs.base_nef import BaseNeuralField


class NeuralSDF(BaseNeuralField):
    def __init__(self, grid: BLASGrid, pos_embedder: str = 'positional', pos_multires: int = 4,
                 position_input: bool = True, activation_type: str = 'relu', layer_type: str = 'none',
                 hidden_dim: int = 128, num_layers: int = 1):
        """grid: occupancy + features (OctreeGrid for NGLOD).  pos_embedder: 'none' | 'identity' | 'positional' with
        `pos_multires` octaves; position_input: also hand the raw position to the decoder.  The remaining arguments
        shape the decoder."""
        super().__init__()
        self.grid = grid
        for k, v in dict(pos_multires=pos_multires, position_input=position_input, activation_type=activation_type,
                         layer_type=layer_type, hidden_dim=hidden_dim, num_layers=num_layers).items():
            setattr(self, k, v)
        self.pos_embedder, self.pos_embed_dim = self.init_embedder(pos_embedder, pos_multires, position_input)
        self.decoder = self.init_decoder(activation_type, layer_type, num_layers, hidden_dim)

    # -- construction hooks kept for subclasses that override them (reference :84-113)
    def init_embedder(self, embedder_type, frequencies=None, position_input=True):
        # neural_sdf.py:97 passes a keyword its embedder factory does not take; what it means (keep the raw input when
        # position_input is set) is what make_position_embedder does
        return _grid_mlp.make_position_embedder(embedder_type, frequencies, position_input)

    def init_decoder(self, activation_type, layer_type, num_layers, hidden_dim):
        return _grid_mlp.make_decoder(self.decoder_input_dim(), 1, activation_type, layer_type, num_layers, hidden_dim)

    def register_forward_functions(self):
        self._register_forward_function(self.sdf, ["sdf"])

    def sdf(self, coords, lod_idx=None):
        """Signed distance at `coords` ([batch, 3] or [batch, num_samples, 3]); the output keeps the leading shape."""
        lead = coords.shape[:-1]
        if coords.shape[0] == 0:
            return dict(sdf=coords.new_zeros(*lead, 1))
        lod_idx = self.grid.num_lods - 1 if lod_idx is None else lod_idx
        out = _grid_mlp.decode(self.grid, self.decoder, self.pos_embedder, coords.reshape(-1, 3), lod_idx, embed_first=True)
        return dict(sdf=out.reshape(*lead, 1))

    def effective_feature_dim(self):
        return _grid_mlp.grid_feature_width(self.grid)

    def decoder_input_dim(self):
        return self.effective_feature_dim() + (self.pos_embed_dim if self.position_input else 0)

    def public_properties(self) -> Dict[str, Any]:
        return {"Grid": self.grid, "Pos. Embedding": self.pos_embedder, "Decoder (sdf)": self.decoder}
