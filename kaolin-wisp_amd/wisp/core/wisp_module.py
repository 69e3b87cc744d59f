"""WispModule: common base of grids, fields, tracers and acceleration structures.
Mirror of wisp/core/wisp_module.py:14-40 (an nn.Module that can name itself and list public properties)."""
from abc import ABC, abstractmethod
from typing import Any, Dict

import torch.nn as nn


class WispModule(nn.Module, ABC):
    def __init__(self):
        super().__init__()

    def name(self) -> str:
        """Human readable name; the class name unless overridden."""
        return type(self).__name__

    @abstractmethod
    def public_properties(self) -> Dict[str, Any]:
        """Table of outward-facing attributes (logging / GUI)."""
        raise NotImplementedError('Wisp modules should implement the `public_properties` method')
