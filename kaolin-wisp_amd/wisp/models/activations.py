"""Activation factory (subset of wisp/models/activations: only what the in-scope configs select)."""
import torch
import torch.nn.functional as F


def get_activation_class(activation_type):
    """'none' -> identity, 'relu' -> F.relu, 'sin' -> torch.sin.  Exotic sorters (fullsort / minmax) are out of scope."""
    if activation_type == 'none':
        return lambda x: x
    if activation_type == 'relu':
        return F.relu
    if activation_type == 'sin':
        return torch.sin
    raise NotImplementedError(f"activation '{activation_type}' is not provided by this backend")
