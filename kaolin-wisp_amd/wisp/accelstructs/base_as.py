"""Acceleration-structure interface and result containers.
Field names / order and method signatures follow wisp/accelstructs/base_as.py:17-167 exactly (they are
constructed positionally and by keyword elsewhere); bodies are this project's."""
from __future__ import annotations

from abc import ABC
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch

from wisp.core import WispModule


@dataclass
class ASQueryResults:
    """Result of query(): cell indices of the structure for every input coordinate."""
    pidx: torch.LongTensor
    """[num_coords] cell index or -1; [num_coords, level+1] with the parent chain when with_parents=True."""


@dataclass
class ASRaytraceResults:
    """Result of raytrace(): every ray / cell intersection ("nugget"), ordered by ray then front to back."""
    ridx: torch.LongTensor
    """[num_nuggets] index of the ray of each nugget."""
    pidx: torch.LongTensor
    """[num_nuggets] index of the intersected cell in the point hierarchy."""
    depth: torch.FloatTensor
    """[num_nuggets, 1] entry depth, or [num_nuggets, 2] (entry, exit)."""


@dataclass
class ASRaymarchResults:
    """Result of raymarch(): packed samples along the rays, ordered by ray then by depth."""
    samples: torch.FloatTensor
    """[num_hit_samples, 3] sample coordinates."""
    ridx: torch.LongTensor
    """[num_hit_samples] ray index of every sample."""
    depth_samples: Optional[torch.FloatTensor]
    """[num_hit_samples, 1] depth of every sample."""
    deltas: Optional[torch.FloatTensor]
    """[num_hit_samples, 1] distance to the previous sample along the ray."""
    boundary: Optional[torch.BoolTensor]
    """[num_hit_samples] True at the first sample of every ray's pack."""
    pack_info: Optional[torch.IntTensor] = None
    """start index of every pack (== boundary.nonzero()); filled by the HIP raymarch kernels."""


class BaseAS(WispModule, ABC):
    """Interface of all acceleration structures."""

    def __init__(self):
        super().__init__()

    def query(self, coords, level=None, with_parents=False) -> ASQueryResults:
        raise NotImplementedError(f"{self.name} acceleration structure does not support the 'query' method.")

    def raytrace(self, rays, level=None, with_exit=False) -> ASRaytraceResults:
        raise NotImplementedError(f"{self.name} acceleration structure does not support the 'raytrace' method.")

    def raymarch(self, rays, *args, **kwargs) -> ASRaymarchResults:
        raise NotImplementedError(f"{self.name} acceleration structure does not support the 'raymarch' method.")

    def occupancy(self) -> List[int]:
        return list()

    def capacity(self) -> List[int]:
        return list()

    def public_properties(self) -> Dict[str, Any]:
        return {'#Used Cells (LOD)': self.occupancy(), '#Capacity (LOD)': self.capacity()}
