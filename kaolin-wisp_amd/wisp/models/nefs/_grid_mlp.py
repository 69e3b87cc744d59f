"""Shared plumbing for the small "grid features (+) embedded position -> MLP" fields (NeuralSDF, ImageNeuralField).

The reference repeats this construction in each nef; here it is one helper so that the two fields are only a channel
name, an output width and an output activation on top of it.
"""
import torch

from wisp.models.activations import get_activation_class
from wisp.models.decoders import BasicDecoder
from wisp.models.embedders import get_positional_embedder
from wisp.models.layers import get_layer_class


def grid_feature_width(grid):
    """Width of grid.interpolate(...)'s last dimension: LODs concatenated ('cat') or summed."""
    if grid.multiscale_type == 'cat':
        return grid.feature_dim * grid.num_lods
    return grid.feature_dim


def make_position_embedder(kind, frequencies, include_input, coord_dim=3):
    """-> (module or None, width).  kind: 'none' | 'identity' | 'positional'."""
    if kind == 'positional':
        emb, _ = get_positional_embedder(frequencies=frequencies, include_input=include_input)
        return emb, (coord_dim if include_input else 0) + 2 * frequencies * coord_dim
    if kind == 'identity' or (kind == 'none' and include_input):
        return torch.nn.Identity(), coord_dim
    if kind == 'none':
        return None, 0
    raise NotImplementedError(f'Unsupported embedder type: {kind}')


def make_decoder(in_width, out_width, activation_type, layer_type, num_layers, hidden_dim):
    return BasicDecoder(input_dim=in_width, output_dim=out_width, activation=get_activation_class(activation_type),
                        bias=True, layer=get_layer_class(layer_type), num_layers=num_layers, hidden_dim=hidden_dim,
                        skip=[])


def decode(grid, decoder, embedder, coords, lod_idx, embed_first):
    """coords [N, D] -> decoder output [N, out].  Features and embedding are concatenated in the order the checkpoint
    layout of the corresponding reference field expects (embedding first for the SDF, features first for images)."""
    n = coords.shape[0]
    feats = grid.interpolate(coords[:, None] if coords.shape[-1] == 3 else coords, lod_idx).reshape(n, -1)
    if embedder is not None:
        emb = embedder(coords).reshape(n, -1)
        feats = torch.cat([emb, feats] if embed_first else [feats, emb], dim=-1)
    return decoder(feats)
