"""Oracle (test infrastructure): Structured-Point-Cloud octree build, point query and ray/octree
intersection in numpy.

Restates the Kaolin-Core 0.13 leaves that wisp's acceleration structure calls (source not vendored,
see oracle/__init__.py), at the call sites:
  * wisp/ops/spc/conversions.py:15-48,72-88  (pointcloud_to_octree, octree_to_spc)
  * wisp/ops/spc/constructors.py:14-28       (create_dense_octree)
  * wisp/accelstructs/octree_as.py:146-186   (query, raytrace)
Data model follows SURVEY.md Appendix A.1.  Integer outputs are the bit-exact contract of the HIP
kernels; float depths are defined by the float32 operation order written in `slab_test`.

Parity: the compositions built on the leaves - pointcloud_to_octree, dilate_points, create_dense_octree, octree_to_spc - are PINNED to the
reference's function bodies (tests/test_reference_modules.py, tests/golden/spc_builders_ref.npz).  The Kaolin leaves themselves
(points <-> morton, quantize_points, points_to_octree, scan / generate_points, query, raytrace, make_dual / make_trinkets) are
UNPINNED: Kaolin's source is not available here; they are checked against hand-computed cases, float64 brute force and structural
properties (tests/test_oracle_golden.py).
"""
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- morton / quantise
def points_to_morton(points):
    """int points [N,3] -> int64 morton codes; per bit i: z -> 3i, y -> 3i+1, x -> 3i+2 (A.1)."""
    p = np.asarray(points).astype(np.int64)
    code = np.zeros(p.shape[0], dtype=np.int64)
    for i in range(16):
        code |= ((p[:, 0] >> i) & 1) << (3 * i + 2)
        code |= ((p[:, 1] >> i) & 1) << (3 * i + 1)
        code |= ((p[:, 2] >> i) & 1) << (3 * i + 0)
    return code


def morton_to_points(codes):
    c = np.asarray(codes).astype(np.int64)
    p = np.zeros((c.shape[0], 3), dtype=np.int64)
    for i in range(16):
        p[:, 0] |= ((c >> (3 * i + 2)) & 1) << i
        p[:, 1] |= ((c >> (3 * i + 1)) & 1) << i
        p[:, 2] |= ((c >> (3 * i + 0)) & 1) << i
    return p.astype(np.int16)


def quantize_points(x, level):
    """kaolin quantize_points (conversions.py:29): clamp(floor(2^level*(0.5x+0.5)), 0, 2^level-1)."""
    x = np.asarray(x, dtype=F32)
    res = F32(2 ** level)
    q = np.floor(res * (F32(0.5) * x + F32(0.5)))
    return np.clip(q, 0, 2 ** level - 1).astype(np.int16)


# ----------------------------------------------------------------------------- build
def points_to_octree(points, level):
    """kaolin unbatched_points_to_octree(points, level, sorted=False) (octree_as.py:132):
    unique -> morton sort -> one occupancy byte per non-leaf node, breadth first."""
    m = np.unique(points_to_morton(points))
    per_level = []
    for _ in range(level):
        parents = m >> 3
        uniq, inv = np.unique(parents, return_inverse=True)
        bytes_ = np.zeros(uniq.shape[0], dtype=np.uint8)
        np.bitwise_or.at(bytes_, inv, (1 << (m & 7)).astype(np.uint8))
        per_level.append(bytes_)
        m = uniq
    if level == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.concatenate(per_level[::-1])


_POPC = np.array([bin(i).count("1") for i in range(256)], dtype=np.int32)


def scan_octree(octree):
    """kaolin scan_octrees for one octree (conversions.py:85): returns (max_level, pyramid[2,L+2] int64,
    exsum int32[len+1])."""
    octree = np.asarray(octree, dtype=np.uint8)
    pc = _POPC[octree]
    exsum = np.zeros(octree.shape[0] + 1, dtype=np.int32)
    np.cumsum(pc, out=exsum[1:])
    counts = [1]
    pos = 0
    while pos < octree.shape[0]:
        n = counts[-1]
        counts.append(int(pc[pos:pos + n].sum()))
        pos += n
    level = len(counts) - 1
    pyramid = np.zeros((2, level + 2), dtype=np.int64)
    pyramid[0, :level + 1] = counts
    pyramid[1, 1:] = np.cumsum(pyramid[0, :-1])
    return level, pyramid, exsum


def generate_points(octree, pyramid, exsum):
    """kaolin generate_points (conversions.py:86): int16 point hierarchy, all levels concatenated."""
    level = pyramid.shape[1] - 2
    total = int(pyramid[1, -1])
    pts = np.zeros((total, 3), dtype=np.int16)
    for l in range(level):
        s, n = int(pyramid[1, l]), int(pyramid[0, l])
        if n == 0:
            continue
        bits = octree[s:s + n]
        parent = pts[s:s + n].astype(np.int32)
        for c in range(8):
            has = (bits >> c) & 1 == 1
            if not has.any():
                continue
            rank = _POPC[bits & ((2 << c) - 1 & 0xFF)]
            child_idx = exsum[s:s + n] + rank
            off = np.array([(c >> 2) & 1, (c >> 1) & 1, c & 1], dtype=np.int32)
            pts[child_idx[has]] = (2 * parent[has] + off).astype(np.int16)
    return pts


def octree_to_spc(octree):
    """wisp octree_to_spc (conversions.py:72-88) -> (points, pyramid, exsum)."""
    _, pyramid, exsum = scan_octree(octree)
    return generate_points(octree, pyramid, exsum), pyramid, exsum


def create_dense_octree(level):
    """wisp create_dense_octree (constructors.py:14-28): every cell of `level` occupied."""
    n = sum(8 ** l for l in range(level))
    return np.full(n, 255, dtype=np.uint8)


# processing.py:26-41, duplicates removed: everything in {-1,0,1}^3 except the centre and the three two-negative edges
DILATE_OFFSETS = tuple(o for o in ((x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1))
                       if o not in ((0, 0, 0), (-1, -1, 0), (-1, 0, -1), (0, -1, -1)))


def dilate_points(points, level):
    """wisp dilate_points (ops/spc/processing.py:13-47): the offsets the reference's list spells out - 6 faces, 8 corners and 9 of the
    12 edges: it has no -x-y, -x-z, -y-z term (`+y-x` etc. cover the mixed signs, nothing covers two negatives) and no `points` term
    either, so a cell grows into 23 neighbours without itself - clipped to the grid, unique, morton order."""
    p = np.asarray(points).astype(np.int64)
    offs = np.array([o for o in DILATE_OFFSETS], dtype=np.int64)
    grown = np.clip((p[None, :, :] + offs[:, None, :]).reshape(-1, 3), 0, 2 ** level - 1)
    return morton_to_points(np.unique(points_to_morton(grown)))


def pointcloud_to_octree(pointcloud, level, attributes=None, dilate=0):
    """wisp pointcloud_to_octree (conversions.py:15-48): quantise, dilate `dilate` times, unique, morton sort, octree; with
    `attributes` [N, F] also the per-cell mean (float32 sum in input order / count) in morton order.  (The reference allocates the
    accumulator as zeros_like(unique): it only runs for F == 3; the restatement takes any F.)"""
    points = quantize_points(pointcloud, level)
    for _ in range(dilate):
        points = dilate_points(points, level)
    codes = points_to_morton(points)
    morton, inverse, counts = np.unique(codes, return_inverse=True, return_counts=True)
    octree = points_to_octree(morton_to_points(morton), level)
    if attributes is None:
        return octree
    a = np.asarray(attributes, dtype=F32)
    att = np.zeros((morton.shape[0], a.shape[1]), dtype=F32)
    for i in range(a.shape[0]):                                   # index_add_ on the CPU adds in input order
        att[inverse[i]] += a[i]
    return octree, (att / counts[:, None].astype(F32)).astype(F32)


# ----------------------------------------------------------------------------- query
def query(octree, exsum, coords, level, with_parents=False):
    """kaolin unbatched_query (octree_as.py:162).

    Oracle definition (upstream float path unverifiable): a point with any |x| > 1 (or NaN) is outside
    -> -1.  Otherwise q = min(floor(2^level * fl32(0.5*x + 0.5)), 2^level - 1) in float32, then walk
    root -> level with child = xbit<<2 | ybit<<1 | zbit.
    Returns int64 [Q] (leaf index or -1) or [Q, level+1] with parents.
    """
    coords = np.asarray(coords, dtype=F32).reshape(-1, 3)
    Q = coords.shape[0]
    inside = np.all(np.abs(coords) <= F32(1.0), axis=1)
    res = 2 ** level
    with np.errstate(invalid="ignore"):
        q = np.floor(F32(res) * (F32(0.5) * coords + F32(0.5)))
    q = np.where(np.isfinite(q), q, 0)
    q = np.minimum(q, res - 1).astype(np.int64)
    out = np.full((Q, level + 1), -1, dtype=np.int64)
    node = np.zeros(Q, dtype=np.int64)
    alive = inside.copy()
    out[alive, 0] = 0
    for l in range(level):
        sh = level - 1 - l
        c = (((q[:, 0] >> sh) & 1) << 2) | (((q[:, 1] >> sh) & 1) << 1) | ((q[:, 2] >> sh) & 1)
        bits = octree[np.where(alive, node, 0)].astype(np.int64)
        has = ((bits >> c) & 1) == 1
        alive = alive & has
        nxt = exsum[np.where(alive, node, 0)] + _POPC[(bits & ((2 << c) - 1)) & 0xFF]
        node = np.where(alive, nxt, 0)
        out[alive, l + 1] = node[alive]
    return out if with_parents else out[:, level].copy()


# ----------------------------------------------------------------------------- raytrace
def slab_test(o, inv, pts, level):
    """Ray vs axis-aligned cell of `level` (float32, every op rounded separately, no FMA).

    cell centre c = r*(2p+1) - 1, r = 2^-level (exact in float32); lo = c - r, hi = c + r;
    t0 = (lo - o) * inv, t1 = (hi - o) * inv, inv = 1/d (IEEE division; d = 0 gives +-inf);
    near = fmin(t0,t1), far = fmax(t0,t1) (NaN-ignoring); tmin = max over axes of near,
    tmax = min over axes of far; entry = fmax(tmin, 0); hit iff tmax > entry.
    Returns (hit, entry, exit=tmax).
    """
    r = F32(1.0 / (1 << level))
    c = r * (F32(2.0) * pts.astype(F32) + F32(1.0)) - F32(1.0)
    lo = c - r
    hi = c + r
    with np.errstate(invalid="ignore", over="ignore"):
        t0 = (lo - o) * inv
        t1 = (hi - o) * inv
    near = np.fmin(t0, t1)
    far = np.fmax(t0, t1)
    tmin = np.fmax(np.fmax(near[:, 0], near[:, 1]), near[:, 2])
    tmax = np.fmin(np.fmin(far[:, 0], far[:, 1]), far[:, 2])
    entry = np.fmax(tmin, F32(0.0))
    with np.errstate(invalid="ignore"):
        hit = tmax > entry
    return hit, entry.astype(F32), tmax.astype(F32), c


def raytrace(octree, points, pyramid, exsum, origins, dirs, level, with_exit=False):
    """kaolin unbatched_raytrace(..., return_depth=True, with_exit) (octree_as.py:183-185).

    Level-by-level "decide / subdivide" traversal: every (ray, node) nugget is slab-tested at its own
    level; survivors of a non-target level are replaced by their existing children, visited in the order
    child = i XOR code, i = 0..7, where code has bit (4,2,1) set iff the ray ORIGIN lies on the positive
    (x,y,z) side of the node centre (closest octant first => front-to-back for exact arithmetic).
    Rays' dist_min/dist_max are not used (they are not passed at the call site).
    Returns ridx int32[M], pidx int32[M], depth float32 [M,1] (entry) or [M,2] (entry, exit).
    """
    o = np.asarray(origins, dtype=F32).reshape(-1, 3)
    d = np.asarray(dirs, dtype=F32).reshape(-1, 3)
    with np.errstate(divide="ignore"):
        inv_all = (F32(1.0) / d).astype(F32)
    ridx = np.arange(o.shape[0], dtype=np.int64)
    pidx = np.zeros(o.shape[0], dtype=np.int64)
    for l in range(level + 1):
        hit, entry, exit_, c = slab_test(o[ridx], inv_all[ridx], points[pidx], l)
        ridx, pidx, entry, exit_, c = ridx[hit], pidx[hit], entry[hit], exit_[hit], c[hit]
        if l == level:
            break
        oo = o[ridx]
        code = ((oo[:, 0] > c[:, 0]).astype(np.int64) << 2) | ((oo[:, 1] > c[:, 1]).astype(np.int64) << 1) \
            | (oo[:, 2] > c[:, 2]).astype(np.int64)
        bits = octree[pidx].astype(np.int64)
        base = exsum[pidx].astype(np.int64)
        # expand: for i in 0..7 child j = i ^ code; keep order (nugget-major, i-minor)
        i = np.arange(8, dtype=np.int64)[None, :]
        j = i ^ code[:, None]
        has = ((bits[:, None] >> j) & 1) == 1
        child = base[:, None] + _POPC[(bits[:, None] & ((2 << j) - 1)) & 0xFF]
        rr = np.broadcast_to(ridx[:, None], has.shape)
        ridx, pidx = rr[has], child[has]
    depth = np.stack([entry, exit_], axis=1) if with_exit else entry[:, None]
    return ridx.astype(np.int32), pidx.astype(np.int32), depth.astype(F32)


# ----------------------------------------------------------------------------- pack utilities (A.4)
def mark_pack_boundaries(ids):
    """kaolin mark_pack_boundaries (octree_as.py:300): b[0]=True, b[i] = ids[i] != ids[i-1]."""
    ids = np.asarray(ids)
    b = np.ones(ids.shape[0], dtype=bool)
    if ids.shape[0] > 1:
        b[1:] = ids[1:] != ids[:-1]
    return b


def inclusive_sum(x):
    """kaolin._C.render.spc.inclusive_sum_cuda (octree_as.py:351): int32 inclusive scan."""
    return np.cumsum(np.asarray(x, dtype=np.int32), dtype=np.int64).astype(np.int32)


# ----------------------------------------------------------------------------- dual octree / trinkets (A.1)
def make_dual(points, pyramid):
    """kaolin unbatched_make_dual (constructors.py:45): per level, the unique integer corners
    p + {0,1}^3 of all voxels, morton-sorted.  Returns (points_dual int16, pyramid_dual int64[2,L+2])."""
    level = pyramid.shape[1] - 2
    duals, counts = [], []
    for l in range(level + 1):
        s, n = int(pyramid[1, l]), int(pyramid[0, l])
        p = points[s:s + n].astype(np.int64)
        corners = (p[:, None, :] + _CORNER_OFFS[None, :, :]).reshape(-1, 3)
        m = np.unique(points_to_morton(corners))
        duals.append(morton_to_points(m))
        counts.append(m.shape[0])
    pyr = np.zeros((2, level + 2), dtype=np.int64)
    pyr[0, :level + 1] = counts
    pyr[1, 1:] = np.cumsum(pyr[0, :-1])
    return np.concatenate(duals).astype(np.int16), pyr


_CORNER_OFFS = np.array([[(j >> 2) & 1, (j >> 1) & 1, j & 1] for j in range(8)], dtype=np.int64)


def make_trinkets(points, pyramid, points_dual, pyramid_dual):
    """kaolin unbatched_make_trinkets (constructors.py:46): trinkets[p, j] = index, local to the dual
    block of p's level, of corner j = dx<<2|dy<<1|dz; parents[p] = parent point index (-1 for root)."""
    level = pyramid.shape[1] - 2
    total = int(pyramid[1, -1])
    trinkets = np.zeros((total, 8), dtype=np.int32)
    parents = np.full(total, -1, dtype=np.int32)
    for l in range(level + 1):
        s, n = int(pyramid[1, l]), int(pyramid[0, l])
        ds, dn = int(pyramid_dual[1, l]), int(pyramid_dual[0, l])
        dm = points_to_morton(points_dual[ds:ds + dn])
        p = points[s:s + n].astype(np.int64)
        corners = (p[:, None, :] + _CORNER_OFFS[None, :, :]).reshape(-1, 3)
        idx = np.searchsorted(dm, points_to_morton(corners))
        trinkets[s:s + n] = idx.reshape(n, 8).astype(np.int32)
        if l > 0:
            ps = int(pyramid[1, l - 1])
            pm = points_to_morton(points[ps:ps + int(pyramid[0, l - 1])])
            parents[s:s + n] = (ps + np.searchsorted(pm, points_to_morton(points[s:s + n]) >> 3)).astype(np.int32)
    return trinkets, parents
