"""Positional (Fourier) embedding of coordinates / view directions
(wisp/models/embedders/positional_embedder.py:18-100; output layout [x ; sin(2^k x) k-major ; cos(same)])."""
from typing import Any, Dict

import torch
import torch.nn as nn

from wisp.core import WispModule


class PositionalEmbedder(WispModule):
    def __init__(self, num_freq, max_freq_log2, log_sampling=True, include_input=True, input_dim=3):
        super().__init__()
        self.num_freq = num_freq
        self.max_freq_log2 = max_freq_log2
        self.log_sampling = log_sampling
        self.include_input = include_input
        if log_sampling:
            bands = 2.0 ** torch.linspace(0.0, max_freq_log2, steps=num_freq)
        else:
            bands = torch.linspace(1, 2.0 ** max_freq_log2, steps=num_freq)
        self.out_dim = (input_dim if include_input else 0) + bands.shape[0] * input_dim * 2
        self.bands = nn.Parameter(bands).requires_grad_(False)

    def forward(self, coords):
        """coords [N, input_dim] -> [N, out_dim]."""
        n = coords.shape[0]
        winded = (coords[:, None] * self.bands[None, :, None]).reshape(n, coords.shape[1] * self.num_freq)
        parts = [torch.sin(winded), torch.cos(winded)]
        if self.include_input:
            parts.insert(0, coords)
        return torch.cat(parts, dim=-1)

    def name(self) -> str:
        return "Positional Encoding"

    def public_properties(self) -> Dict[str, Any]:
        return {"Output Dim": self.out_dim, "Num. Frequencies": self.num_freq,
                "Max Frequency": f"2^{self.max_freq_log2}", "Include Input": self.include_input}


def get_positional_embedder(frequencies, input_dim=3, include_input=True):
    """(embedder, out_dim) for bands 2^0 .. 2^(frequencies-1)."""
    encoder = PositionalEmbedder(frequencies, frequencies - 1, input_dim=input_dim, include_input=include_input)
    return encoder, encoder.out_dim
