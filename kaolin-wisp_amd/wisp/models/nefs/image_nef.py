"""ImageNeuralField (app/image): pixel coordinates -> hash-grid features (+) 3-octave embedding -> MLP -> sigmoid RGB.
Mirrors wisp/models/nefs/image_nef.py:35-97; `rgb()` hands back the tensor itself because ImageTrainer consumes it that
way (wisp/trainers/image_trainer.py:66)."""
import torch

from wisp.models.grids import BLASGrid
from wisp.models.nefs import _grid_mlp
from wisp.models.nefs.base_nef import BaseNeuralField

_OCTAVES = 3


class ImageNeuralField(BaseNeuralField):
    def __init__(self, grid: BLASGrid, activation_type: str = 'relu', layer_type: str = 'none', hidden_dim: int = 128,
                 num_layers: int = 1):
        super().__init__()
        self.grid = grid
        self.activation_type, self.layer_type = activation_type, layer_type
        self.hidden_dim, self.num_layers = hidden_dim, num_layers
        n_lods = len(grid.resolutions)
        self.feature_dim = grid.feature_dim * (n_lods if grid.multiscale_type == 'cat' else 1)
        self.embedder, self.embed_dim = _grid_mlp.make_position_embedder('positional', _OCTAVES, True, coord_dim=2)
        self.input_dim = self.feature_dim + self.embed_dim
        self.decoder = _grid_mlp.make_decoder(self.input_dim, 3, activation_type, layer_type, num_layers, hidden_dim)

    def register_forward_functions(self):
        self._register_forward_function(self.rgb, ["rgb"])

    def rgb(self, coords, lod=None):
        """coords [batch, 2] in [-1, 1] -> colours [batch, 3] in (0, 1)."""
        lod = len(self.grid.resolutions) - 1 if lod is None else lod
        return torch.sigmoid(_grid_mlp.decode(self.grid, self.decoder, self.embedder, coords, lod, embed_first=False))
