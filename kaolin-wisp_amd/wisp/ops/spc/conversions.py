"""Structured-point-cloud (SPC) octree construction.

Host-side build code in torch tensor ops (runs at construction time and once per prune, not per step).
Provides what wisp/ops/spc/conversions.py:15-88 obtains from Kaolin-Core (quantize_points, points_to_morton,
morton_to_points, unbatched_points_to_octree, scan_octrees, generate_points, unbatched_get_level_points);
data model: SURVEY.md Appendix A.1.
"""
import torch


def default_device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


_POPC_TABLE = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32)


def popcount_u8(x):
    return _POPC_TABLE.to(x.device)[x.long()]


def points_to_morton(points):
    """int [N,3] -> int64 morton; per bit i: z -> 3i, y -> 3i+1, x -> 3i+2."""
    p = points.long()
    code = torch.zeros(p.shape[0], dtype=torch.int64, device=p.device)
    for i in range(16):
        code |= ((p[:, 0] >> i) & 1) << (3 * i + 2)
        code |= ((p[:, 1] >> i) & 1) << (3 * i + 1)
        code |= ((p[:, 2] >> i) & 1) << (3 * i)
    return code


def morton_to_points(codes):
    c = codes.long()
    p = torch.zeros(c.shape[0], 3, dtype=torch.int64, device=c.device)
    for i in range(16):
        p[:, 0] |= ((c >> (3 * i + 2)) & 1) << i
        p[:, 1] |= ((c >> (3 * i + 1)) & 1) << i
        p[:, 2] |= ((c >> (3 * i)) & 1) << i
    return p.short()


def quantize_points(x, level):
    """float coords in [-1,1] -> int16 cell coords of `level` (clamped)."""
    res = 2 ** level
    q = torch.floor(res * (0.5 * x.float() + 0.5))
    return torch.clamp(q, 0, res - 1).short()


def unbatched_points_to_octree(points, level, sorted=False):
    """Quantised points [N,3] -> occupancy bytes of every non-leaf node, BFS / morton order."""
    built = build_spc(level, points=points) if points.shape[0] else None
    if built is not None:
        octree = built[0]
        # OctreeAS(octree) picks the finished hierarchy up - if the bytes are still the ones it was derived from
        octree._wisp_spc_parts = built[1:] + (octree._version,)
        return octree
    m = points_to_morton(points)
    m = torch.unique(m)                     # sorted + deduplicated
    per_level = []
    for _ in range(level):
        parents = m >> 3
        uniq, inv = torch.unique_consecutive(parents, return_inverse=True)
        byte = torch.zeros(uniq.shape[0], dtype=torch.int32, device=m.device)
        byte.index_add_(0, inv, (1 << (m & 7)).int())      # children are distinct, so the sum is the OR
        per_level.append(byte.to(torch.uint8))
        m = uniq
    if not per_level:
        return torch.zeros(0, dtype=torch.uint8, device=points.device)
    return torch.cat(per_level[::-1])


def build_spc(level, points=None, leaf_mask=None):
    """(octree, points, pyramid, exsum) in one go on the GPU (csrc/spc.hip, no sort), or None when the inputs are not on a
    GPU / the level is beyond the dense-mask build / nothing is occupied - callers then use the generic path below."""
    src = points if points is not None else leaf_mask
    if not (torch.is_tensor(src) and src.is_cuda):
        return None
    import wisp._C as _C
    if level < 1 or level > _C.SPC_DEVICE_BUILD_MAX_LEVEL:
        return None
    if points is not None and points.dtype != torch.int16:
        points = points.to(torch.int16)
    return _C.spc_build(level, points=points, leaf_mask=leaf_mask)


def scan_octrees(octree):
    """-> (max_level, pyramid int32 [2, L+2] on the CPU, exsum int32 [len+1])."""
    pc = popcount_u8(octree)
    exsum = torch.zeros(octree.shape[0] + 1, dtype=torch.int32, device=octree.device)
    exsum[1:] = torch.cumsum(pc, 0)
    ex = exsum.cpu()
    counts, pos, total = [1], 0, octree.shape[0]
    while pos < total:
        n = counts[-1]
        counts.append(int(ex[pos + n] - ex[pos]))
        pos += n
    level = len(counts) - 1
    pyramid = torch.zeros(2, level + 2, dtype=torch.int32)
    pyramid[0, :level + 1] = torch.tensor(counts, dtype=torch.int32)
    pyramid[1, 1:] = torch.cumsum(pyramid[0, :-1], 0)
    return level, pyramid, exsum


def generate_points(octree, pyramid, exsum):
    """int16 point hierarchy [sum_l P_l, 3]; child = 2*parent + (xbit, ybit, zbit)."""
    level = pyramid.shape[1] - 2
    total = int(pyramid[1, -1])
    dev = octree.device
    pts = torch.zeros(total, 3, dtype=torch.int16, device=dev)
    for l in range(level):
        s, n = int(pyramid[1, l]), int(pyramid[0, l])
        if n == 0:
            continue
        bits = octree[s:s + n].int()
        parent = pts[s:s + n].int()
        base = exsum[s:s + n]
        for c in range(8):
            has = ((bits >> c) & 1) == 1
            if not bool(has.any()):
                continue
            rank = popcount_u8((bits & ((2 << c) - 1) & 0xFF).to(torch.uint8))
            child = (base + rank)[has].long()
            off = torch.tensor([(c >> 2) & 1, (c >> 1) & 1, c & 1], dtype=torch.int32, device=dev)
            pts[child] = (2 * parent[has] + off).short()
    return pts


def unbatched_get_level_points(points, pyramid, level):
    s, n = int(pyramid[1, level]), int(pyramid[0, level])
    return points[s:s + n]


def octree_to_spc(octree):
    """(points, pyramid, exsum) for one octree (wisp/ops/spc/conversions.py:72-88)."""
    _, pyramid, exsum = scan_octrees(octree)
    return generate_points(octree, pyramid, exsum), pyramid, exsum


def pointcloud_to_octree(pointcloud, level, attributes=None, dilate=0):
    """Float coordinates in [-1,1] -> octree (wisp/ops/spc/conversions.py:15-48).  With `attributes`,
    also returns the per-voxel mean of the attributes in morton order."""
    from .processing import dilate_points
    points = quantize_points(pointcloud.contiguous().to(default_device()), level)
    for _ in range(dilate):
        points = dilate_points(points, level)
    morton_all = points_to_morton(points)
    morton, inverse, counts = torch.unique(morton_all, return_inverse=True, return_counts=True)
    octree = unbatched_points_to_octree(morton_to_points(morton), level, sorted=True)
    if attributes is None:
        return octree
    att = torch.zeros(morton.shape[0], attributes.shape[1], dtype=torch.float32, device=morton.device)
    att = att.index_add_(0, inverse, attributes.float().to(morton.device)) / counts[:, None].float()
    return octree, att


def mesh_to_octree(vertices, faces, level, num_samples=100000000, chunk=1 << 24):
    """Octree of the `level` cells touched by a triangle mesh (wisp/ops/spc/conversions.py:91-109): area-weighted surface
    samples plus a copy jittered by half a cell, quantised.  The reference materialises all 2 x num_samples points and
    sorts them; here the draw runs in chunks that only set bits of a dense cell mask (2^(3 level) bytes: 2 MiB at level 7),
    so 10^8 samples need 0.4 GB of scratch instead of 5 GB.  Not deterministic (it samples), like the reference."""
    from wisp.ops import mesh as mesh_ops
    dev = default_device()
    V, F = vertices.to(dev), faces.to(dev)
    res = 2 ** level
    mask = torch.zeros(res * res * res, dtype=torch.bool, device=dev)
    distrib = mesh_ops.area_weighted_distribution(V, F)
    done = 0
    while done < num_samples:
        n = min(chunk, num_samples - done)
        pts = mesh_ops.sample_surface(V, F, n, distrib)[0]
        pts = torch.cat([pts, pts + (torch.rand_like(pts) * 2.0 - 1.0) * (1.0 / (2 ** (level + 1)))], dim=0)
        q = quantize_points(pts, level).long()
        mask[(q[:, 0] * res + q[:, 1]) * res + q[:, 2]] = True
        done += n
    cells = mask.nonzero()[:, 0]
    pts = torch.stack([cells // (res * res), (cells // res) % res, cells % res], dim=1).short()
    return unbatched_points_to_octree(pts, level)
