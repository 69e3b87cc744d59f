"""Rays: a pack of ray origins / directions plus the near / far distances.
Boundary value type, API-identical to wisp/core/rays.py:19-198."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple, Union

import torch

INFINITY = torch.finfo().max


def _slice_bound(bound, idx):
    return bound[idx] if isinstance(bound, torch.Tensor) else bound


@dataclass
class Rays:
    origins: torch.Tensor
    """ray origins, shape [..., 3]"""
    dirs: torch.Tensor
    """ray directions, shape [..., 3]"""
    dist_min: Union[float, torch.Tensor] = 0.0
    """distance at which marching starts (near plane)"""
    dist_max: Union[float, torch.Tensor] = INFINITY
    """distance at which marching stops (far plane)"""

    def __len__(self) -> int:
        if self.origins.shape != self.dirs.shape:
            raise Exception(f"Rays.origins shape should match Rays.dirs shape, but got "
                            f"{self.origins.shape} and {self.dirs.shape}.")
        return self.origins.shape[0]

    @property
    def shape(self) -> Tuple[...]:
        return self.origins.shape[:-1]

    @property
    def ndim(self) -> int:
        return self.origins.ndim - 1

    @staticmethod
    def _merge(rays_list, op, dim):
        return Rays(origins=op([r.origins for r in rays_list], dim=dim),
                    dirs=op([r.dirs for r in rays_list], dim=dim),
                    dist_min=min(r.dist_min for r in rays_list),
                    dist_max=max(r.dist_max for r in rays_list))

    @classmethod
    def cat(cls, rays_list: List[Rays], dim: int = 0) -> Rays:
        if dim < 0:
            dim -= 1    # the trailing xyz axis is not a spatial dimension
        nd = rays_list[0].ndim
        if dim > nd - 1 or dim < -nd:
            raise IndexError(f"Dimension out of range (expected to be in range of [{-nd}, {nd - 1}, but got {dim})")
        return cls._merge(rays_list, torch.cat, dim)

    @classmethod
    def stack(cls, rays_list: List[Rays], dim: int = 0) -> Rays:
        return cls._merge(rays_list, torch.stack, dim)

    def __getitem__(self, idx) -> Rays:
        return Rays(self.origins[idx], self.dirs[idx], _slice_bound(self.dist_min, idx), _slice_bound(self.dist_max, idx))

    def split(self, split_size) -> List[Rays]:
        pairs = zip(torch.split(self.origins, split_size), torch.split(self.dirs, split_size))
        return [Rays(o, d, dist_min=self.dist_min, dist_max=self.dist_max) for o, d in pairs]

    def reshape(self, *dims: Tuple) -> Rays:
        def _b(x):
            return x.reshape(*dims[:-1]) if torch.is_tensor(x) else x
        return Rays(self.origins.reshape(*dims), self.dirs.reshape(*dims), _b(self.dist_min), _b(self.dist_max))

    def squeeze(self, dim: int) -> Rays:
        return Rays(self.origins.squeeze(dim), self.dirs.squeeze(dim), self.dist_min, self.dist_max)

    def contiguous(self) -> Rays:
        return Rays(self.origins.contiguous(), self.dirs.contiguous(), self.dist_min, self.dist_max)

    def to(self, *args, **kwargs) -> Rays:
        o, d = self.origins.to(*args, **kwargs), self.dirs.to(*args, **kwargs)
        if o is self.origins and d is self.dirs:
            return self
        return Rays(o, d, self.dist_min, self.dist_max)
