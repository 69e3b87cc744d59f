"""SynLego: the procedural stand-in for NeRF-synthetic "Lego" (no dataset exists on disk).

An analytic density + colour field inside [-1,1]^3 (a studded brick assembly), cameras on the upper hemisphere
with the Lego intrinsics (800x800, camera_angle_x = 0.6911112, radius 4.0311/1.25, near/far 1/5, black
background - wisp/datasets/formats/nerf_standard_dataset.py:394-403), and ground-truth colours rendered by
dense quadrature of the analytic field.  Everything is plain torch and runs on whatever device it is given;
it feeds bench.py, __graft_entry__.smoke() and the tests - it is input generation, not part of the hot path.
"""
import math

import numpy as np
import torch

CAMERA_ANGLE_X = 0.6911112
CAMERA_RADIUS = 4.0311 / 1.25
NEAR, FAR = 1.0, 5.0
SIGMA = 60.0

# (centre, half-extent) boxes of the brick assembly
_BOXES = [((0.0, -0.25, 0.0), (0.62, 0.10, 0.42)),
          ((-0.25, -0.02, 0.0), (0.30, 0.13, 0.30)),
          ((0.30, 0.03, -0.12), (0.20, 0.18, 0.20)),
          ((0.05, 0.28, 0.05), (0.12, 0.14, 0.12))]
# studs: (centre x, z, top-surface y, radius, height)
_STUDS = [(x, z, -0.15, 0.06, 0.05) for x in (-0.45, -0.15, 0.15, 0.45) for z in (-0.28, 0.28)] + \
         [(-0.25, 0.0, 0.11, 0.07, 0.05), (0.30, -0.12, 0.21, 0.07, 0.05)]


def density(x):
    """sigma(x) [..]: SIGMA inside the assembly, 0 outside."""
    inside = torch.zeros(x.shape[:-1], dtype=torch.bool, device=x.device)
    for c, h in _BOXES:
        c_t = torch.tensor(c, device=x.device, dtype=x.dtype)
        h_t = torch.tensor(h, device=x.device, dtype=x.dtype)
        inside |= ((x - c_t).abs() <= h_t).all(-1)
    for cx, cz, y0, r, hgt in _STUDS:
        rad = (x[..., 0] - cx) ** 2 + (x[..., 2] - cz) ** 2 <= r * r
        inside |= rad & (x[..., 1] >= y0) & (x[..., 1] <= y0 + hgt)
    sphere = ((x - torch.tensor((-0.45, 0.25, 0.25), device=x.device, dtype=x.dtype)) ** 2).sum(-1) <= 0.15 ** 2
    inside |= sphere
    return inside.to(x.dtype) * SIGMA


def colour(x):
    """view-independent albedo in [0,1]^3."""
    ph = torch.tensor((0.0, 2.1, 4.2), device=x.device, dtype=x.dtype)
    k = torch.tensor((5.0, 7.0, 3.0), device=x.device, dtype=x.dtype)
    return 0.5 + 0.5 * torch.sin(x * k + ph + 2.0 * x.roll(1, -1))


def cameras(num_views, seed=0):
    """camera-to-world rotation [V,3,3] and position [V,3]; upper hemisphere, looking at the origin."""
    rng = np.random.default_rng(seed)
    theta = rng.uniform(0, 2 * np.pi, num_views)
    phi = np.arccos(rng.uniform(0.05, 0.95, num_views))          # elevation from +y
    pos = CAMERA_RADIUS * np.stack([np.sin(phi) * np.cos(theta), np.cos(phi), np.sin(phi) * np.sin(theta)], 1)
    fwd = -pos / np.linalg.norm(pos, axis=1, keepdims=True)
    up = np.tile(np.array([0.0, 1.0, 0.0]), (num_views, 1))
    right = np.cross(fwd, up); right /= np.linalg.norm(right, axis=1, keepdims=True)
    true_up = np.cross(right, fwd)
    rot = np.stack([right, true_up, fwd], 2)                    # columns: right, up, forward
    return torch.from_numpy(rot.astype(np.float32)), torch.from_numpy(pos.astype(np.float32))


def pixel_rays(rot, pos, view_idx, px, py, res=800):
    """pinhole rays through pixel centres (wisp/ops/raygen/raygen.py:40-85 convention: unit directions)."""
    focal = 0.5 * res / math.tan(0.5 * CAMERA_ANGLE_X)
    x = (px.float() + 0.5 - res / 2) / focal
    y = -(py.float() + 0.5 - res / 2) / focal
    d_cam = torch.stack([x, y, torch.ones_like(x)], -1)
    d = torch.einsum('nij,nj->ni', rot[view_idx], d_cam)
    d = torch.nn.functional.normalize(d, dim=-1)
    return pos[view_idx].contiguous(), d.contiguous()


@torch.no_grad()
def render_gt(origins, dirs, steps=768, chunk=1 << 16):
    """ground-truth rgb [N,3] by dense quadrature of the analytic field, black background."""
    out = []
    for s in range(0, origins.shape[0], chunk):
        o, d = origins[s:s + chunk], dirs[s:s + chunk]
        t = torch.linspace(NEAR, FAR, steps + 1, device=o.device)
        tm = 0.5 * (t[1:] + t[:-1])
        x = o[:, None, :] + d[:, None, :] * tm[None, :, None]
        tau = density(x) * (t[1] - t[0])
        T = torch.exp(-(torch.cumsum(tau, 1) - tau))
        w = T * (1 - torch.exp(-tau))
        out.append((w[..., None] * colour(x)).sum(1))
    return torch.cat(out, 0)


def ray_bank(num_rays, num_views=100, res=800, seed=0, device='cpu', with_gt=True):
    """A bank of random training rays: (origins, dirs, rgb) - the on-device equivalent of SampleRays over the
    [V, H*W, 3] tensors of NeRFSyntheticDataset."""
    rot, pos = cameras(num_views, seed)
    g = torch.Generator().manual_seed(seed + 1)
    v = torch.randint(0, num_views, (num_rays,), generator=g)
    px = torch.randint(0, res, (num_rays,), generator=g)
    py = torch.randint(0, res, (num_rays,), generator=g)
    o, d = pixel_rays(rot.to(device), pos.to(device), v.to(device), px.to(device), py.to(device), res)
    rgb = render_gt(o, d) if with_gt else None
    return o, d, rgb


def occupied_cells(level, device='cpu', probes=3):
    """int16 [P,3] cells of `level` that contain density (the steady state a pruned BLAS converges to)."""
    res = 2 ** level
    idx = torch.arange(res, device=device)
    cells = torch.stack(torch.meshgrid(idx, idx, idx, indexing='ij'), -1).reshape(-1, 3)
    occ = torch.zeros(cells.shape[0], dtype=torch.bool, device=device)
    offs = torch.linspace(0.0, 1.0, probes, device=device)
    for ox in offs:
        for oy in offs:
            for oz in offs:
                p = (cells.float() + torch.stack([ox, oy, oz])) / res * 2 - 1
                occ |= density(p) > 0
    return cells[occ].short()


# ---------------------------------------------------------------------------------------------------------------------
# Stand-ins for the other BASELINE.json configs: none of their datasets exists on disk either.
@torch.no_grad()
def first_hit_depth(origins, dirs, steps=1024, chunk=1 << 15):
    """depth [N] of the first quadrature node inside the SynLego density (inf for rays that miss) - the stand-in for the
    RTMV depth maps that `OctreeAS.from_pointcloud` is fed from (wisp/datasets/formats/rtmv_dataset.py:517-570)."""
    out = []
    for s in range(0, origins.shape[0], chunk):
        o, d = origins[s:s + chunk], dirs[s:s + chunk]
        t = torch.linspace(NEAR, FAR, steps, device=o.device)
        inside = density(o[:, None, :] + d[:, None, :] * t[None, :, None]) > 0
        first = torch.where(inside.any(1), inside.float().argmax(1), torch.full((o.shape[0],), -1, device=o.device))
        out.append(torch.where(first >= 0, t[first.clamp(min=0)], torch.full_like(t[first.clamp(min=0)], float('inf'))))
    return torch.cat(out)


def v8_pointcloud(num_rays=1 << 20, num_views=100, res=400, seed=0, device='cpu'):
    """SynV8: surface point cloud of the scene from `num_views` mip-2-sized (400x400) depth maps -> [M,3] in [-1,1]."""
    o, d, _ = ray_bank(num_rays, num_views=num_views, res=res, seed=seed, device=device, with_gt=False)
    t = first_hit_depth(o, d)
    keep = torch.isfinite(t)
    return (o[keep] + d[keep] * t[keep, None]).contiguous()


def render_gt_white(origins, dirs, steps=768):
    """SynV8 / VQAD ground truth: the same field composited over a WHITE background (tests/apps/test_nerf.py:74-76)."""
    out = []
    for s in range(0, origins.shape[0], 1 << 16):
        o, d = origins[s:s + (1 << 16)], dirs[s:s + (1 << 16)]
        t = torch.linspace(NEAR, FAR, steps + 1, device=o.device)
        tm = 0.5 * (t[1:] + t[:-1])
        x = o[:, None, :] + d[:, None, :] * tm[None, :, None]
        tau = density(x) * (t[1] - t[0])
        T = torch.exp(-(torch.cumsum(tau, 1) - tau))
        w = T * (1 - torch.exp(-tau))
        out.append((w[..., None] * colour(x)).sum(1) + (1 - w.sum(1, keepdim=True)))
    return torch.cat(out, 0)


# SynArmadillo: an analytic signed-distance "creature" - smooth union of spheres and capsules inside the unit sphere.
_SDF_SPHERES = [((0.0, 0.05, 0.0), 0.34), ((0.0, 0.50, 0.04), 0.20), ((0.10, 0.66, 0.10), 0.07), ((-0.10, 0.66, 0.10), 0.07)]
_SDF_CAPSULES = [((0.22, 0.18, 0.0), (0.55, 0.30, 0.10), 0.08), ((-0.22, 0.18, 0.0), (-0.55, 0.30, 0.10), 0.08),
                 ((0.14, -0.22, 0.0), (0.22, -0.70, 0.05), 0.10), ((-0.14, -0.22, 0.0), (-0.22, -0.70, 0.05), 0.10),
                 ((0.0, -0.05, -0.25), (0.0, -0.35, -0.60), 0.06)]


def armadillo_sdf(x, k=0.08):
    """signed distance [..] (negative inside) of the SynArmadillo body; polynomial smooth minimum with radius k."""
    def smin(a, b):
        h = torch.clamp(0.5 + 0.5 * (b - a) / k, 0.0, 1.0)
        return b + (a - b) * h - k * h * (1.0 - h)
    dist = None
    for c, r in _SDF_SPHERES:
        s = (x - torch.tensor(c, device=x.device, dtype=x.dtype)).norm(dim=-1) - r
        dist = s if dist is None else smin(dist, s)
    for a, b, r in _SDF_CAPSULES:
        a_t, b_t = torch.tensor(a, device=x.device, dtype=x.dtype), torch.tensor(b, device=x.device, dtype=x.dtype)
        pa, ba = x - a_t, b_t - a_t
        h = torch.clamp((pa * ba).sum(-1) / (ba * ba).sum(), 0.0, 1.0)
        dist = smin(dist, (pa - ba * h[..., None]).norm(dim=-1) - r)
    return dist


def armadillo_surface_points(num=500000, seed=0, device='cpu', iters=12):
    """points on the zero level set: uniform draws in the bounding box projected along the SDF gradient (finite
    differences) - the stand-in for 'sample the mesh surface' (nglod_octree.yaml:54: 500 000 samples per epoch)."""
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(num, 3, generator=g) * 1.8 - 0.9).to(device)
    eps = 1e-3
    e = torch.eye(3, device=device) * eps
    for _ in range(iters):
        dist = armadillo_sdf(x)
        grad = torch.stack([armadillo_sdf(x + e[i]) - armadillo_sdf(x - e[i]) for i in range(3)], -1) / (2 * eps)
        x = x - dist[..., None] * torch.nn.functional.normalize(grad, dim=-1)
    keep = armadillo_sdf(x).abs() < 2e-3
    return x[keep].contiguous()


def armadillo_training_samples(num, seed=0, device='cpu'):
    """(coords [num,3], sdf [num,1]) in the mix of SDFDataset's sample modes (surface / near-surface / uniform thirds,
    wisp/datasets/sdf_dataset.py 'sample_mode')."""
    g = torch.Generator().manual_seed(seed + 17)
    third = num // 3
    surf = armadillo_surface_points(third * 2 + 1024, seed=seed, device=device)[:third * 2]
    near = surf[third:] + torch.randn(surf[third:].shape, generator=g).to(device) * 0.01
    uni = (torch.rand(num - 2 * third, 3, generator=g) * 2 - 1).to(device)
    x = torch.cat([surf[:third], near, uni], 0)
    return x.contiguous(), armadillo_sdf(x)[..., None].contiguous()
