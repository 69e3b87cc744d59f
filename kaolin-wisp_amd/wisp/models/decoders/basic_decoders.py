"""BasicDecoder: the small MLP used by every neural field (interface of
wisp/models/decoders/basic_decoders.py:16-150; parameter names `layers.<i>` / `lout` are kept for checkpoints)."""
from typing import Any, Dict

import torch
import torch.nn as nn

from wisp.core import WispModule


class BasicDecoder(WispModule):
    def __init__(self, input_dim, output_dim, activation, bias, layer=nn.Linear, num_layers=1, hidden_dim=128, skip=[]):
        """
        Args:
            input_dim / output_dim (int): MLP input / output widths.
            activation (callable): hidden activation.
            bias (bool): use biases.
            layer (nn.Module class): linear layer class.
            num_layers (int): number of hidden layers.
            hidden_dim (int): hidden width.
            skip (List[int]): hidden layers after which the input is concatenated back.
        """
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.activation = activation
        self.bias = bias
        self.layer = layer
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.skip = [] if skip is None else skip
        self.make()

    def make(self):
        widths = []
        for i in range(self.num_layers):
            if i == 0:
                widths.append(self.input_dim)
            elif i in self.skip:
                widths.append(self.hidden_dim + self.input_dim)
            else:
                widths.append(self.hidden_dim)
        self.layers = nn.ModuleList([self.layer(w, self.hidden_dim, bias=self.bias) for w in widths])
        self.lout = self.layer(self.hidden_dim, self.output_dim, bias=self.bias)

    def forward(self, x, return_h=False):
        """x [batch, ..., input_dim] -> [batch, ..., output_dim] (and the last hidden activations if return_h)."""
        h = x
        for i, l in enumerate(self.layers):
            h = self.activation(l(h))
            if i != 0 and i in self.skip:
                h = torch.cat([x, h], dim=-1)
        out = self.lout(h)
        return (out, h) if return_h else out

    def initialize(self, get_weight):
        """Re-initialise every weight matrix with get_weight(matrix)."""
        for l in list(self.layers) + [self.lout]:
            l.weight = nn.Parameter(get_weight(l.weight))

    def name(self) -> str:
        return "BasicDecoder"

    def public_properties(self) -> Dict[str, Any]:
        return {"Input Dim": self.input_dim, "Hidden Dim": self.hidden_dim, "Output Dim": self.output_dim,
                "Num. Layers": self.num_layers, "Layer Type": self.layer.__name__,
                "Activation": getattr(self.activation, '__name__', str(self.activation)),
                "Bias": self.bias, "Skip Connections": self.skip}
