"""Finite-difference gradients of scalar fields (wisp/ops/differential/gradients.py:29-45)."""
import torch


def finitediff_gradient(x, f, eps=0.005):
    """Central differences of f: R^3 -> R at x [..., 3]."""
    offs = torch.eye(3, device=x.device) * eps
    parts = [f(x + offs[a]) - f(x - offs[a]) for a in range(3)]
    return torch.cat(parts, dim=-1) / (eps * 2.0)


def autodiff_gradient(x, f):
    """Gradient of f at x through autograd, with the graph kept so that it can be differentiated again
    (wisp/ops/differential/gradients.py:14-26)."""
    with torch.enable_grad():
        x = x.requires_grad_(True)
        y = f(x)
        return torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y), create_graph=True)[0]


def tetrahedron_gradient(x, f, eps=0.005):
    """Four-point gradient estimate: f sampled at the corners (+,-,-), (-,-,+), (-,+,-), (+,+,+) of a tetrahedron of half-width eps
    around x, each sample weighted by its corner's signs (gradients.py:48-95)."""
    corners = torch.tensor([[1.0, -1.0, -1.0], [-1.0, -1.0, 1.0], [-1.0, 1.0, -1.0], [1.0, 1.0, 1.0]], device=x.device)
    total = None
    for k in corners:
        term = k * f((x + k * eps).detach())
        total = term if total is None else total + term
    return total / (eps * 4.0)
