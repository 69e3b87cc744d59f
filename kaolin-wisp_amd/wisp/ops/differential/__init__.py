"""Finite-difference gradients of scalar fields (wisp/ops/differential/gradients.py:29-45)."""
import torch


def finitediff_gradient(x, f, eps=0.005):
    """Central differences of f: R^3 -> R at x [..., 3]."""
    offs = torch.eye(3, device=x.device) * eps
    parts = [f(x + offs[a]) - f(x - offs[a]) for a in range(3)]
    return torch.cat(parts, dim=-1) / (eps * 2.0)
