"""Mesh helpers behind OctreeAS.from_mesh (wisp/accelstructs/octree_as.py:65-106): OBJ geometry loading, sphere / aabb
normalisation and area-weighted surface sampling - the subset of wisp/ops/mesh the occupancy build uses
(load_obj.py:52, normalize.py:11, per_face_normals.py:11, area_weighted_distribution.py:12, random_face.py:13,
sample_surface.py:13).  Construction-time torch code on whatever device the vertices live on; nothing here is on the
per-step path.  Materials / textures (tinyobjloader in the reference, a feature its own docstring calls unused) are not
read."""
import torch


def load_obj(fname: str, load_materials: bool = False):
    """Vertices float32 [V,3] and triangle indices int64 [F,3] of a Wavefront OBJ (polygons are fan-triangulated, negative
    indices are relative to the vertices read so far, as the format defines)."""
    if load_materials:
        raise NotImplementedError("load_obj(load_materials=True): textures / materials are not read by this backend "
                                  "(OctreeAS.from_mesh(sample_tex=True) is documented as unused in the reference)")
    verts, faces = [], []
    with open(fname) as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    if not verts or not faces:
        raise ValueError(f"{fname}: no geometry (need 'v' and 'f' records)")
    return torch.tensor(verts, dtype=torch.float32), torch.tensor(faces, dtype=torch.int64)


def normalize(V: torch.Tensor, F: torch.Tensor, mode: str):
    """'sphere': centre the bounding box and scale the farthest vertex onto the unit sphere; 'aabb': scale the bounding
    box into [-1, 1] by its longest side."""
    if mode == 'sphere':
        centre = (V.max(dim=0)[0] + V.min(dim=0)[0]) / 2.0
        V = V - centre
        return V * (1.0 / torch.sqrt((V ** 2).sum(-1).max())), F
    if mode == 'aabb':
        V = V - V.min(dim=0)[0]
        return V * (1.0 / V.max()) * 2.0 - 1.0, F
    if mode == 'planar':                    # x and z fill [-1, 1] on their own, y keeps the common scale and starts at 0
        V = V - V.min(dim=0)[0]
        V[..., 0] *= 1.0 / V[..., 0].max()
        V[..., 2] *= 1.0 / V[..., 2].max()
        V[..., 1] *= 1.0 / V.max()
        V = V * 2.0 - 1.0
        V[..., 1] -= V[..., 1].min()
        return V, F
    if mode == 'none':
        return V, F
    raise ValueError(f"normalize: unsupported mode {mode!r}")


def per_face_normals(V: torch.Tensor, F: torch.Tensor):
    """Unnormalised face normals [F,3] (their length is twice the triangle area)."""
    tri = V[F]
    return torch.cross(tri[:, 0] - tri[:, 1], tri[:, 1] - tri[:, 2], dim=1)     # the reference's edge pair (per_face_normals.py:24-27)


def area_weighted_distribution(V: torch.Tensor, F: torch.Tensor, normals: torch.Tensor = None):
    """Categorical distribution over the faces, proportional to their area."""
    if normals is None:
        normals = per_face_normals(V, F)
    areas = torch.norm(normals, p=2, dim=1) * 0.5
    return torch.distributions.Categorical(areas / (areas.sum() + 1e-10))


def random_face(V: torch.Tensor, F: torch.Tensor, num_samples: int, distrib=None):
    """(faces [N,3], their normals [N,3]) drawn area-weighted."""
    if distrib is None:
        distrib = area_weighted_distribution(V, F)
    idx = distrib.sample([num_samples])
    return F[idx], per_face_normals(V, F)[idx]


def sample_surface(V: torch.Tensor, F: torch.Tensor, num_samples: int, distrib=None):
    """(points [N,3] uniformly distributed over the surface, normals [N,3]); barycentric draw (1-sqrt(r1), sqrt(r1)(1-r2),
    sqrt(r1) r2)."""
    fidx, normals = random_face(V, F, num_samples, distrib)
    tri = V[fidx]
    u = torch.sqrt(torch.rand(num_samples, 1, device=V.device))
    v = torch.rand(num_samples, 1, device=V.device)
    return (1 - u) * tri[:, 0] + (u * (1 - v)) * tri[:, 1] + (u * v) * tri[:, 2], normals


def sample_near_surface(V: torch.Tensor, F: torch.Tensor, num_samples: int, variance: float = 0.01, distrib=None):
    """Surface samples pushed off the surface by gaussian noise of standard deviation `variance`
    (wisp/ops/mesh/sample_near_surface.py:13-34)."""
    if distrib is None:
        distrib = area_weighted_distribution(V, F)
    samples = sample_surface(V, F, num_samples, distrib)[0]
    return samples + torch.randn_like(samples) * variance


def sample_uniform(num_samples: int):
    """Uniform samples in [-1, 1]^3, on the host like the reference (sample_uniform.py:11-19)."""
    return torch.rand(num_samples, 3) * 2.0 - 1.0


def point_sample(V: torch.Tensor, F: torch.Tensor, techniques: list, num_samples: int):
    """`num_samples` points per entry of `techniques`, concatenated in order: 'trace' = on the surface, 'near' = near it,
    'rand' = uniform in the cube; unknown names are skipped (point_sample.py:15-48)."""
    distrib = area_weighted_distribution(V, F) if ('trace' in techniques or 'near' in techniques) else None
    parts = []
    for technique in techniques:
        if technique == 'trace':
            parts.append(sample_surface(V, F, num_samples, distrib=distrib)[0])
        elif technique == 'near':
            parts.append(sample_near_surface(V, F, num_samples, distrib=distrib))
        elif technique == 'rand':
            parts.append(sample_uniform(num_samples).to(V.device))
    return torch.cat(parts, dim=0)


def barycentric_coordinates(points: torch.Tensor, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor):
    """Barycentric coordinates [N,3] of `points` in the triangles (A, B, C), each clipped to [0, 1]
    (barycentric_coordinates.py:11-45)."""
    e0, e1, rel = B - A, C - A, points - A
    d00, d01, d11 = (e0 * e0).sum(-1), (e0 * e1).sum(-1), (e1 * e1).sum(-1)
    d20, d21 = (rel * e0).sum(-1), (rel * e1).sum(-1)
    denom = d00 * d11 - d01 * d01
    out = torch.zeros(points.shape[0], 3, device=points.device)
    out[..., 1] = torch.clip((d11 * d20 - d01 * d21) / denom, 0.0, 1.0)
    out[..., 2] = torch.clip((d00 * d21 - d01 * d20) / denom, 0.0, 1.0)
    out[..., 0] = torch.clip(1.0 - (out[..., 1] + out[..., 2]), 0.0, 1.0)
    return out
