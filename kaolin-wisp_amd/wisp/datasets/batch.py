"""Batch containers exchanged between datasets and trainers (shape contract of wisp/datasets/batch.py:19-110)."""
from typing import Any, Dict, List, Optional

import torch

from wisp.core import Rays


class Batch(dict):
    """dict with a `fields` view; the exact channels are up to the dataset."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @property
    def fields(self) -> List[str]:
        return list(self.keys())


class MultiviewBatch(Batch):
    """rays (wisp.core.Rays) + optional cameras + optional per-ray supervision channels (rgb by convention)."""

    def __init__(self, rays: Rays, cameras: Optional[list] = None, rgb: Optional[torch.Tensor] = None, *args, **kwargs):
        super().__init__(rays=rays, cameras=cameras, rgb=rgb)
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def ray_values(self) -> Dict[str, Any]:
        """per-ray channels other than the rays themselves."""
        skip = ("rays", "cameras")
        return {k: v for k, v in self.items() if k not in skip and v is not None}


class SDFBatch(Batch):
    """coords [N,3] + signed distance sdf [N,1] (+ optional extra channels)."""

    def __init__(self, coords: torch.Tensor, sdf: torch.Tensor, *args, **kwargs):
        super().__init__(coords=coords, sdf=sdf)
        for k, v in dict(*args, **kwargs).items():
            self[k] = v
