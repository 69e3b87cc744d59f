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


class _SmallDecoder(torch.autograd.Function):
    """BasicDecoder with one hidden relu layer, biases and a single output as one HIP launch per direction
    (wisp_small_decoder_fwd / _bwd); what NeuralSDF's decoder is in every shipped NGLOD config."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        import wisp._C as C
        x, w1, b1, w2, b2 = (t.detach().float().contiguous() for t in (x, w1, b1, w2, b2))
        w2 = w2.reshape(-1)
        ctx.save_for_backward(x, w1, b1, w2, b2)
        return C.small_decoder_forward(x, w1, b1, w2, b2)

    @staticmethod
    def backward(ctx, grad_out):
        import wisp._C as C
        x, w1, b1, w2, b2 = ctx.saved_tensors
        gx, gw1, gb1, gw2, gb2 = C.small_decoder_backward(x, w1, b1, w2, b2, grad_out.contiguous().float())
        return gx, gw1, gb1, gw2.reshape(1, -1), gb2


def _fusable_small_decoder(decoder, feats):
    import os
    if os.environ.get("WISP_SMALL_DECODER_FUSED", "1") == "0" or not feats.is_cuda or feats.shape[-1] > 32:
        return False
    layers = getattr(decoder, 'layers', None)
    lout = getattr(decoder, 'lout', None)
    return (type(decoder) is BasicDecoder and layers is not None and len(layers) == 1 and not decoder.skip
            and type(layers[0]) is torch.nn.Linear and type(lout) is torch.nn.Linear and layers[0].bias is not None
            and lout.bias is not None and lout.out_features == 1 and layers[0].out_features <= 256
            and decoder.activation in (torch.relu, torch.nn.functional.relu) and layers[0].weight.dtype == torch.float32)


def decode(grid, decoder, embedder, coords, lod_idx, embed_first):
    """coords [N, D] -> decoder output [N, out].  Features and embedding are concatenated in the order the checkpoint
    layout of the corresponding reference field expects (embedding first for the SDF, features first for images)."""
    n = coords.shape[0]
    feats = grid.interpolate(coords[:, None] if coords.shape[-1] == 3 else coords, lod_idx).reshape(n, -1)
    if embedder is not None:
        emb = embedder(coords).reshape(n, -1)
        feats = torch.cat([emb, feats] if embed_first else [feats, emb], dim=-1)
    if _fusable_small_decoder(decoder, feats):
        return _SmallDecoder.apply(feats, decoder.layers[0].weight, decoder.layers[0].bias, decoder.lout.weight, decoder.lout.bias)
    return decoder(feats)
