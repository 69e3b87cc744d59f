"""RenderBuffer: named per-ray output channels produced by a tracer.
Same surface as wisp/core/render_buffer.py:21-439 (rgb / alpha / depth first-class, arbitrary extra channels,
missing channels read as None, `+` concatenates along the ray axis), implemented as a plain channel dictionary
instead of the reference's per-channel-set dynamic dataclass."""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Set, Tuple

import numpy as np
import torch

_CORE = ("rgb", "alpha", "depth")


class RenderBuffer:
    def __init__(self, rgb=None, alpha=None, depth=None, **custom):
        object.__setattr__(self, "_ch", {"rgb": rgb, "alpha": alpha, "depth": depth, **custom})

    # ---- channel access -------------------------------------------------------------------------
    def __getattr__(self, item):
        ch = object.__getattribute__(self, "_ch")
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return ch.get(item, None)                       # unknown channels read as None (render_buffer.py:92-97)

    def __setattr__(self, key, value):
        self._ch[key] = value

    def __iter__(self) -> Iterator[Tuple[str, Optional[torch.Tensor]]]:
        return iter(self._ch.items())

    def __getstate__(self):
        return dict(self._ch)

    def __setstate__(self, state):
        object.__setattr__(self, "_ch", dict(state))

    def __repr__(self):
        body = ", ".join(f"{k}={None if v is None else tuple(v.shape)}" for k, v in self._ch.items())
        return f"RenderBuffer({body})"

    @property
    def rgba(self) -> Optional[torch.Tensor]:
        if self.rgb is None or self.alpha is None:
            return None
        return torch.cat((self.rgb, self.alpha), dim=-1)

    @rgba.setter
    def rgba(self, val):
        self._ch["rgb"] = None if val is None else val[..., 0:-1]
        self._ch["alpha"] = None if val is None else val[..., -1:]

    @property
    def channels(self) -> Set[str]:
        return {k for k, v in self._ch.items() if v is not None}

    def has_channel(self, name: str) -> bool:
        return name in self.channels

    def get_channel(self, name: str) -> Optional[torch.Tensor]:
        return self._ch.get(name, None)

    # ---- element-wise plumbing ------------------------------------------------------------------
    def _apply(self, fn) -> RenderBuffer:
        return RenderBuffer(**{k: (None if v is None else fn(v)) for k, v in self._ch.items()})

    @staticmethod
    def _apply_on_pair(a: RenderBuffer, b: RenderBuffer, fn) -> RenderBuffer:
        keys = list(dict.fromkeys(list(a._ch.keys()) + list(b._ch.keys())))
        return RenderBuffer(**{k: fn((a._ch.get(k), b._ch.get(k))) for k in keys})

    def cat(self, other: RenderBuffer, dim: int = 0) -> RenderBuffer:
        def _cat(pair):
            x, y = pair
            if x is None:
                return y
            if y is None:
                return x
            return torch.cat((x, y), dim=dim)
        return RenderBuffer._apply_on_pair(self, other, _cat)

    def __add__(self, other: RenderBuffer) -> RenderBuffer:
        return self.cat(other, dim=0)

    @staticmethod
    def mean(*rblst) -> RenderBuffer:
        def _sum(pair):
            x, y = pair
            if x is None or y is None:
                return None
            return x.float() + y.float()
        total = rblst[0]
        for rb in rblst[1:]:
            total = RenderBuffer._apply_on_pair(total, rb, _sum)
        return total._apply(lambda x: x / float(len(rblst)))

    def blend(self, other: RenderBuffer, channel_kit=None) -> RenderBuffer:
        """Depth-ordered blend of two buffers (viewer feature; only the default 'closest wins' rule is offered)."""
        if self.depth is None or other.depth is None:
            return self
        closer = (self.depth <= other.depth)
        def _pick(pair):
            x, y = pair
            if x is None or y is None:
                return x if y is None else y
            return torch.where(closer.expand_as(x) if closer.shape[-1] == 1 else closer, x, y)
        return RenderBuffer._apply_on_pair(self, other, _pick)

    def transpose(self) -> RenderBuffer:
        return self._apply(lambda x: x.permute(1, 0, *tuple(range(2, x.ndim))))

    def scale(self, size: Tuple, interpolation='bilinear') -> RenderBuffer:
        def _scale(x):
            assert x.ndim == 3, 'RenderBuffer scale() assumes channels have 2D spatial dimensions.'
            y = x.permute(2, 0, 1)[None].float()
            y = torch.nn.functional.interpolate(y, size=size, mode=interpolation)
            return y[0].permute(1, 2, 0).to(x.dtype)
        return self._apply(_scale)

    def numpy_dict(self) -> Dict[str, np.ndarray]:
        return {k: v.detach().cpu().numpy() for k, v in self._ch.items() if v is not None}

    def exr_dict(self) -> Dict[str, torch.Tensor]:
        out = {}
        for k, v in self.numpy_dict().items():
            out[k] = v
        return out

    def image(self) -> RenderBuffer:
        """Channels normalised for display: floats scaled to [0,255], hit expanded, depth normalised."""
        def _img(name, x):
            x = x.float()
            if name == "depth":
                rng = torch.clamp(x.max() - x.min(), min=1e-8)
                x = (x - x.min()) / rng
            if x.shape[-1] == 1:
                x = x.expand(*x.shape[:-1], 3)
            return torch.clamp(x, 0.0, 1.0) * 255.0
        return RenderBuffer(**{k: (None if v is None else _img(k, v)) for k, v in self._ch.items()})

    def reshape(self, *dims) -> RenderBuffer:
        return self._apply(lambda x: x.reshape(*dims))

    def to(self, *args, **kwargs) -> RenderBuffer:
        return self._apply(lambda x: x.to(*args, **kwargs))

    def cuda(self) -> RenderBuffer:
        return self._apply(lambda x: x.cuda())

    def cpu(self) -> RenderBuffer:
        return self._apply(lambda x: x.cpu())

    def detach(self) -> RenderBuffer:
        return self._apply(lambda x: x.detach())

    def byte(self) -> RenderBuffer:
        return self._apply(lambda x: x.byte())

    def half(self) -> RenderBuffer:
        return self._apply(lambda x: x.half())

    def float(self) -> RenderBuffer:
        return self._apply(lambda x: x.float())

    def double(self) -> RenderBuffer:
        return self._apply(lambda x: x.double())
