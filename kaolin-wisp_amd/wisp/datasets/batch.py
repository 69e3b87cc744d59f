"""Batch containers exchanged between datasets and trainers (shape contract of wisp/datasets/batch.py:19-110)."""
from typing import Any, Dict, List, Optional

import torch

from wisp.core import Rays


class Batch(dict):
    """dict with a `fields` view; the exact channels are up to the dataset.  The reference derives from attrdict.AttrDict
    (batch.py:10,19): fields also read and write as attributes (`batch.rays`); that part of AttrDict is provided here."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __getattr__(self, name):
        if name.startswith('__') or name not in self:
            raise AttributeError(f"'{type(self).__name__}' instance has no attribute '{name}'")
        return self[name]

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        if name not in self:
            raise AttributeError(name)
        del self[name]

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
        """per-ray channels other than the rays themselves.  (The reference keeps `rgb` only and drops any further channel handed
        to the constructor, batch.py:65-72; here extra per-ray channels such as masks survive and are sampled along.)"""
        skip = ("rays", "cameras")
        return {k: v for k, v in self.items() if k not in skip and v is not None}


class SDFBatch(Batch):
    """coords [N,3] + signed distance sdf [N,1], optional rgb / normals of the nearest surface point (batch.py:75-110), plus any
    extra channels."""

    def __init__(self, coords: torch.Tensor, sdf: torch.Tensor, rgb: Optional[torch.Tensor] = None,
                 normals: Optional[torch.Tensor] = None, *args, **kwargs):
        super().__init__(coords=coords, sdf=sdf, rgb=rgb, normals=normals)
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def coord_values(self) -> Dict[str, Any]:
        """per-coordinate supervision channels: sdf always, rgb / normals when present."""
        out = dict(sdf=self['sdf'])
        for name in ('rgb', 'normals'):
            if self[name] is not None:
                out[name] = self[name]
        return out
