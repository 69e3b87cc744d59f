"""Pipeline: a neural field plus the tracer that renders it (wisp/models/pipeline.py:14-53)."""
import torch.nn as nn

from wisp.models.nefs import BaseNeuralField
from wisp.tracers import BaseTracer


class Pipeline(nn.Module):
    def __init__(self, nef: BaseNeuralField, tracer: BaseTracer = None):
        super().__init__()
        self.nef: BaseNeuralField = nef
        self.tracer: BaseTracer = tracer

    def forward(self, *args, **kwargs):
        """tracer(nef, ...) when a tracer is attached, else the field itself."""
        if self.tracer is not None:
            return self.tracer(self.nef, *args, **kwargs)
        return self.nef(*args, **kwargs)
