"""BLASGrid: a feature grid paired with a bottom-level acceleration structure
(interface of wisp/models/grids/blas_grid.py:14-80)."""
from abc import ABC, abstractmethod
from typing import Any, Dict, Set, Type

from wisp.core import WispModule
from wisp.accelstructs import BaseAS, ASQueryResults, ASRaytraceResults, ASRaymarchResults


class BLASGrid(WispModule, ABC):
    def __init__(self, blas: BaseAS):
        super().__init__()
        self.blas = blas
        self.num_lods = 1
        self.active_lods = [0]

    def raymarch(self, *args, **kwargs) -> ASRaymarchResults:
        return self.blas.raymarch(*args, **kwargs)

    def raytrace(self, *args, **kwargs) -> ASRaytraceResults:
        return self.blas.raytrace(*args, **kwargs)

    def query(self, *args, **kwargs) -> ASQueryResults:
        return self.blas.query(*args, **kwargs)

    @abstractmethod
    def interpolate(self, coords, lod_idx):
        """Features at `coords` ([batch, num_samples, 3] or [batch, 3]) for level-of-detail index `lod_idx`."""
        raise NotImplementedError('A BLASGrid should implement the interpolation functionality according to '
                                  'the grid structure.')

    def supported_blas(self) -> Set[Type[BaseAS]]:
        return set()

    def public_properties(self) -> Dict[str, Any]:
        return {"Acceleration Structure": self.blas}
