"""BaseTracer: argument plumbing shared by all tracers (wisp/tracers/base_tracer.py:20-175).
forward() negotiates channels and fills trace() keyword arguments from the tracer's own attributes."""
import inspect
from abc import abstractmethod
from typing import Any, Dict

import torch

from wisp.core import Rays, WispModule


class BaseTracer(WispModule):
    def __init__(self, bg_color=(0.0, 0.0, 0.0)):
        super().__init__()
        self.bg_color = bg_color

    @abstractmethod
    def get_supported_channels(self):
        """Channel names this tracer can output."""
        pass

    @abstractmethod
    def get_required_nef_channels(self):
        """Channel names the neural field must provide."""
        pass

    @abstractmethod
    def trace(self, nef, rays, channels, extra_channels, *args, **kwargs):
        """Render `rays` through `nef`; returns a RenderBuffer."""
        pass

    def forward(self, nef, rays: Rays, channels=None, **kwargs):
        """Trace with channel negotiation.  Any trace() keyword the caller omits is taken from the attribute of the
        same name on the tracer (so PackedRFTracer(num_steps=...) defaults apply; base_tracer.py:136-159)."""
        nef_channels = nef.get_supported_channels()
        missing = self.get_required_nef_channels() - nef_channels
        if missing:
            raise Exception(f"The neural field class {type(nef)} does not output the required channels {missing}.")
        if channels is None:
            requested = set(self.get_supported_channels())
        elif isinstance(channels, str):
            requested = {channels}
        else:
            requested = set(channels)
        extra = requested - self.get_supported_channels()
        unsupported = extra - nef_channels
        if unsupported:
            raise Exception(f"Channels {unsupported} are not supported in the tracer {type(self)} or neural field {type(nef)}.")

        own_args = self._trace_arg_names()
        call = {}
        for a in own_args:
            if a in kwargs:
                call[a] = kwargs[a]
            else:
                default = getattr(self, a, None)
                if default is not None:
                    call[a] = default
        torch.cuda.nvtx.range_push("Tracer.trace")         # (base_tracer.py:160; push / pop: the context-manager form costs a generator
        try:                                               #  round trip per call in the host-bound drop-in loop)
            return self.trace(nef, rays, requested, extra, **call)
        finally:
            torch.cuda.nvtx.range_pop()

    def _trace_arg_names(self):
        """names of trace()'s tracer-specific keyword arguments (introspected once per class, not per call)."""
        cache = BaseTracer._ARG_CACHE
        key = type(self).trace
        if key not in cache:
            base_args = set(inspect.signature(BaseTracer.trace).parameters) - {"self", "args", "kwargs"}
            cache[key] = [a for a in inspect.signature(key).parameters if a not in base_args | {"self", "args", "kwargs"}]
        return cache[key]

    _ARG_CACHE = {}

    def public_properties(self) -> Dict[str, Any]:
        return dict()
