"""Layer factory (subset of wisp/models/layers.py: the normalised-linear variants are out of scope)."""
import torch.nn as nn


def get_layer_class(layer_type):
    if layer_type in ('none', 'linear'):
        return nn.Linear
    raise NotImplementedError(f"layer type '{layer_type}' is not provided by this backend")
