"""MultiTable: the stacked per-level feature tables of a hash grid
(layout identical to wisp/models/grids/utils.py:13-67 so state dicts interchange)."""
from typing import Optional, Tuple

import torch
import torch.nn as nn


class MultiTable(nn.Module):
    def __init__(self, resolutions: Tuple[int, ...], coord_dim: int, feature_dim: int, std: float = 0.01,
                 max_feats: Optional[int] = None):
        """Level l holds min(max_feats, res_l ** coord_dim) rows of `feature_dim` features, initialised N(0, std)."""
        super().__init__()
        self.num_lods = len(resolutions)
        self.max_feats = max_feats
        self.coord_dim = coord_dim
        self.feature_dim = feature_dim
        self.resolutions = torch.tensor([[int(r)] for r in resolutions], dtype=torch.int64)   # stays on the host

        sizes = []
        for r in resolutions:
            n = int(r) ** coord_dim
            sizes.append(min(max_feats, n) if max_feats else n)
        begin = [0]
        for n in sizes:
            begin.append(begin[-1] + n)
        self.register_buffer("begin_idxes", torch.tensor(begin, dtype=torch.int64))
        self.register_buffer("num_feats", torch.tensor(sizes, dtype=torch.int64))
        self.total_feats = begin[-1]
        self.feats = nn.Parameter(torch.randn(self.total_feats, self.feature_dim) * std)

    def get_level(self, idx):
        """The [rows, feature_dim] table of level `idx`."""
        return self.feats[self.begin_idxes[idx]:self.begin_idxes[idx + 1]]
