"""ImageNeuralField: 2-D coordinates -> hash-grid features + positional embedding -> MLP -> sigmoid RGB
(config C1, app/image).  Surface of wisp/models/nefs/image_nef.py:35-97; `rgb()` returns the tensor directly because
that is how ImageTrainer calls it (wisp/trainers/image_trainer.py:66)."""
import torch

from wisp.models.activations import get_activation_class
from wisp.models.decoders import BasicDecoder
from wisp.models.embedders import get_positional_embedder
from wisp.models.grids import BLASGrid
from wisp.models.layers import get_layer_class
from wisp.models.nefs.base_nef import BaseNeuralField


class ImageNeuralField(BaseNeuralField):
    def __init__(self,
                 grid: BLASGrid,
                 activation_type: str = 'relu',
                 layer_type: str = 'none',
                 hidden_dim: int = 128,
                 num_layers: int = 1):
        super().__init__()
        self.grid = grid
        self.activation_type = activation_type
        self.layer_type = layer_type
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        if self.grid.multiscale_type == 'cat':
            self.feature_dim = self.grid.feature_dim * len(self.grid.resolutions)
        else:
            self.feature_dim = self.grid.feature_dim
        self.embedder, _ = get_positional_embedder(frequencies=3, include_input=True)
        self.embed_dim = 14          # 2-D coords: 2 + 2*3*2 (the embedder itself is dimension-agnostic)
        self.input_dim = self.feature_dim + self.embed_dim
        self.decoder = BasicDecoder(self.input_dim, 3, get_activation_class(self.activation_type), True,
                                    layer=get_layer_class(self.layer_type), num_layers=self.num_layers,
                                    hidden_dim=self.hidden_dim, skip=[])

    def register_forward_functions(self):
        self._register_forward_function(self.rgb, ["rgb"])

    def rgb(self, coords, lod=None):
        """coords [batch, 2] in [-1, 1] -> rgb [batch, 3]."""
        if lod is None:
            lod = len(self.grid.resolutions) - 1
        batch, _ = coords.shape
        feats = self.grid.interpolate(coords, lod).reshape(-1, self.feature_dim)
        fpos = torch.cat([feats, self.embedder(coords).view(batch, self.embed_dim)], dim=-1)
        return torch.sigmoid(self.decoder(fpos))
