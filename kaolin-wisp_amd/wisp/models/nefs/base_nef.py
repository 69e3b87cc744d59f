"""BaseNeuralField: channel-name -> forward-function registry shared by all neural fields
(interface of wisp/models/nefs/base_nef.py:18-202)."""
import inspect
from abc import abstractmethod
from typing import Any, Dict

import torch

from wisp.core import WispModule


class BaseNeuralField(WispModule):
    def __init__(self):
        super().__init__()
        self._forward_functions = {}
        self.register_forward_functions()
        self.supported_channels = set(c for chans in self._forward_functions.values() for c in chans)

    @property
    def device(self):
        """Device of the first registered parameter."""
        return next(self.parameters()).device

    def _register_forward_function(self, fn, channels):
        self._forward_functions[fn] = {channels} if isinstance(channels, str) else set(channels)

    @abstractmethod
    def register_forward_functions(self):
        """Call self._register_forward_function(fn, [channels...]) for every output function."""
        pass

    def get_forward_function(self, channel):
        if channel not in self.get_supported_channels():
            raise Exception(f"Channel {channel} is not supported in {self.__class__.__name__}")
        for fn, chans in self._forward_functions.items():
            if channel in chans:
                return lambda *args, **kwargs: fn(*args, **kwargs)[channel]

    def get_supported_channels(self):
        return self.supported_channels

    def prune(self):
        """Fields with an occupancy structure override this; default is a no-op."""
        pass

    def forward(self, channels=None, **kwargs):
        """Evaluate the requested channels.  str -> tensor, list -> list, set / None -> dict."""
        if not (channels is None or isinstance(channels, (str, list, set))):
            raise Exception(f"Channels type invalid, got {type(channels)}."
                            "Make sure your arguments for the nef are provided as keyword arguments.")
        if channels is None:
            wanted = set(self.get_supported_channels())
        elif isinstance(channels, str):
            wanted = {channels}
        else:
            wanted = set(channels)
        unsupported = wanted - self.get_supported_channels()
        if unsupported:
            raise Exception(f"Channels {unsupported} are not supported in {self.__class__.__name__}")

        # functions that cover the most requested channels run first
        order = sorted(((len(chans & wanted), fn) for fn, chans in self._forward_functions.items() if chans & wanted),
                       key=lambda t: t[0], reverse=True)
        results = {}
        for _, fn in order:
            provides = self._forward_functions[fn] & wanted
            wanted = wanted - provides
            if not provides:
                continue
            torch.cuda.nvtx.range_push(f"{fn.__name__}")
            required, optional = self._fn_args(fn)
            call = {}
            for a in required:
                if a not in kwargs:
                    raise Exception(f"Argument {a} not found as input to in {self.__class__.__name__}.{fn.__name__}()")
                call[a] = kwargs[a]
            for a in optional:
                if a in kwargs:
                    call[a] = kwargs[a]
            out = fn(**call)
            for c in provides:
                results[c] = out[c]
            torch.cuda.nvtx.range_pop()

        if isinstance(channels, str):
            return results.get(channels, None)
        if isinstance(channels, list):
            return [results[c] for c in channels]
        return results

    _ARGSPEC_CACHE = {}

    @staticmethod
    def _fn_args(fn):
        """(required, optional) argument names of a registered forward function, introspected once."""
        key = getattr(fn, "__func__", fn)
        hit = BaseNeuralField._ARGSPEC_CACHE.get(key)
        if hit is None:
            spec = inspect.getfullargspec(fn)
            n_opt = len(spec.defaults) if spec.defaults else 0
            names = spec.args[1:]                                   # drop self
            hit = (names[:len(names) - n_opt] if n_opt else names, names[len(names) - n_opt:] if n_opt else [])
            BaseNeuralField._ARGSPEC_CACHE[key] = hit
        return hit

    def public_properties(self) -> Dict[str, Any]:
        return dict()
