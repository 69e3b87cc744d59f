"""Oracle (test infrastructure): the three OctreeAS raymarch modes, numpy float32.

Follows wisp/accelstructs/octree_as.py:188-374, wisp/ops/spc/sampling.py:35-71 and
wisp/csrc/ops/uniform_sample_cuda.cu:18-59.  The reference draws its jitter with torch.rand on the
device and is unseeded; here the jitter tensor is an INPUT so that identical ray batches give
identical sample sets.  Every float32 operation is rounded separately (no FMA); the HIP kernels are
compiled with -ffp-contract=off on these expressions so integer outputs (ridx, boundary, S) match
bit for bit.

Parity: PINNED - the three march modes against OctreeAS._raymarch_ray / _raymarch_voxel / _raymarch_uniform compiled from the
reference file (same samples, order and pack boundaries; values 1e-6), uniform_sample against the reference kernel body built for the
host (tests/golden/uniform_ref.npz, bit-exact) - over the unpinned Kaolin leaves of oracle/spc.py.
"""
import numpy as np
from . import spc

F32 = np.float32


def linspace01(n):
    """torch.linspace(0, 1, n) as the CUDA kernel evaluates it (octree_as.py:272):
    step = fl(1/(n-1)); i < n//2 -> fl(step*i) else fl(1 - fl(step*(n-1-i)))."""
    if n == 1:
        return np.zeros(1, dtype=F32)
    step = F32(1.0) / F32(n - 1)
    i = np.arange(n)
    lo = (step * i.astype(F32)).astype(F32)
    hi = (F32(1.0) - (step * (n - 1 - i).astype(F32)).astype(F32)).astype(F32)
    return np.where(i < n // 2, lo, hi).astype(F32)


def ray_depths(num_rays, num_samples, near, far, jitter):
    """octree_as.py:272-277: depth = (linspace + jitter/N) * (far-near) + near, float32 step by step."""
    lin = linspace01(num_samples)[None, :]
    jit = (np.asarray(jitter, dtype=F32).reshape(num_rays, num_samples) / F32(num_samples)).astype(F32)
    depth = (lin + jit).astype(F32)
    depth = (depth * F32(far - near)).astype(F32)
    depth = (depth + F32(near)).astype(F32)
    return depth


def raymarch_ray(octree, exsum, origins, dirs, near, far, num_samples, level, jitter):
    """OctreeAS._raymarch_ray (octree_as.py:247-309).

    Returns dict(ridx int64[S], samples f32[S,3], depth_samples f32[S,1], deltas f32[S,1],
    boundary bool[S]).  Samples are ordered by ray then by step (row-major nonzero, :288).
    deltas are differences of the UNFILTERED depth row (:290-291 precede the filter at :298).
    """
    o = np.asarray(origins, dtype=F32).reshape(-1, 3)
    d = np.asarray(dirs, dtype=F32).reshape(-1, 3)
    R, N = o.shape[0], num_samples
    depth = ray_depths(R, N, near, far, jitter)
    # addcmul(o, d, depth) = o + fl(d*depth): the product is rounded before the add
    samples = (o[:, None, :] + (d[:, None, :] * depth[:, :, None]).astype(F32)).astype(F32)
    pidx = spc.query(octree, exsum, samples.reshape(-1, 3), level).reshape(R, N)
    mask = pidx > -1
    deltas = np.diff(depth, axis=1, prepend=np.full((R, 1), F32(near), dtype=F32)).astype(F32)
    rr, kk = np.nonzero(mask)
    ridx = rr.astype(np.int64)
    return dict(
        ridx=ridx,
        samples=samples[rr, kk],
        depth_samples=depth[rr, kk][:, None],
        deltas=deltas[rr, kk][:, None],
        boundary=spc.mark_pack_boundaries(ridx),
    )


def raymarch_voxel(octree, points, pyramid, exsum, origins, dirs, num_samples, level, jitter):
    """OctreeAS._raymarch_voxel (octree_as.py:188-245) + sample_from_depth_intervals (sampling.py:35-55)
    + expand_pack_boundary (sampling.py:58-71).  jitter: [M, num_samples] in [0,1), M = #nuggets."""
    o = np.asarray(origins, dtype=F32).reshape(-1, 3)
    d = np.asarray(dirs, dtype=F32).reshape(-1, 3)
    ridx, pidx, depth = spc.raytrace(octree, points, pyramid, exsum, o, d, level, with_exit=True)
    M, N = ridx.shape[0], num_samples
    ridx = ridx.astype(np.int64)
    jit = np.asarray(jitter, dtype=F32).reshape(-1, N)[:M]
    steps = (np.arange(N, dtype=F32)[None, :] + jit).astype(F32)
    steps = (steps * F32(1.0 / N)).astype(F32)
    entry, exit_ = depth[:, 0:1], depth[:, 1:2]
    ds = (entry + ((exit_ - entry).astype(F32) * steps).astype(F32)).astype(F32)          # [M,N]
    deltas = np.diff(ds, axis=1, prepend=entry).astype(F32)
    samples = (o[ridx][:, None, :] + (d[ridx][:, None, :] * ds[:, :, None]).astype(F32)).astype(F32)
    first = spc.mark_pack_boundaries(ridx)
    boundary = np.zeros(M * N, dtype=bool)
    boundary[np.nonzero(first)[0] * N] = True
    return dict(
        ridx=np.repeat(ridx, N),
        samples=samples.reshape(M * N, 3),
        depth_samples=ds.reshape(M * N, 1),
        deltas=deltas.reshape(M * N, 1),
        boundary=boundary,
        nuggets=(ridx, pidx, depth),
    )


def uniform_scale(num_samples):
    """octree_as.py:336-338: step = 2*sqrt(3)/N; scale = ceil(1/step); step = 1/scale."""
    step_size = 2 * np.sqrt(3) / num_samples
    scale = int(np.ceil(1.0 / step_size))
    return scale, 1.0 / float(scale)


def raymarch_uniform(octree, points, pyramid, exsum, origins, dirs, num_samples, level):
    """OctreeAS._raymarch_uniform (octree_as.py:311-374) + uniform_sample_cuda_kernel
    (uniform_sample_cuda.cu:18-59): fixed lattice t = (ceil(scale*entry) + k) / scale clipped to each
    nugget; deterministic."""
    o = np.asarray(origins, dtype=F32).reshape(-1, 3)
    d = np.asarray(dirs, dtype=F32).reshape(-1, 3)
    ridx, pidx, depth = spc.raytrace(octree, points, pyramid, exsum, o, d, level, with_exit=True)
    scale, step_size = uniform_scale(num_samples)
    ia = np.ceil((F32(scale) * depth[:, 0]).astype(F32)).astype(np.int32)
    ib = np.ceil((F32(scale) * depth[:, 1]).astype(F32)).astype(np.int32)
    cnt = ib - ia
    keep = cnt != 0
    f_ridx, f_depth, f_cnt = ridx[keep], depth[keep], cnt[keep]
    insum = spc.inclusive_sum(f_cnt)
    out = uniform_sample(scale, f_ridx, f_depth, insum)
    S = out["ridx"].shape[0]
    deltas = np.full((S, 1), F32(step_size), dtype=F32)
    r = out["ridx"]
    samples = (o[r] + (d[r] * out["depth_samples"]).astype(F32)).astype(F32)
    return dict(ridx=r, samples=samples, depth_samples=out["depth_samples"], deltas=deltas,
                boundary=out["boundary"], nuggets=(ridx, pidx, depth))


def uniform_sample(scale, ridx, depth, insum):
    """uniform_sample_cuda_kernel (uniform_sample_cuda.cu:18-59): nugget i emits
    n = insum[i]-insum[i-1] samples at depth inv_scale*(ceil(scale*entry)+k); boundary marks the first
    sample of each run of equal ridx."""
    V = ridx.shape[0]
    total = int(insum[-1]) if V else 0
    o_ridx = np.zeros(total, dtype=np.int64)
    o_depth = np.zeros((total, 1), dtype=F32)
    o_bound = np.zeros(total, dtype=bool)
    inv_scale = F32(1.0) / F32(scale)
    starts = np.concatenate([[0], insum[:-1]]).astype(np.int64) if V else np.zeros(0, np.int64)
    first = spc.mark_pack_boundaries(ridx) if V else np.zeros(0, bool)
    for i in range(V):
        n = int(insum[i]) - int(starts[i])
        base = np.ceil(F32(scale) * depth[i, 0]).astype(F32)
        k = np.arange(n, dtype=F32)
        o_ridx[starts[i]:starts[i] + n] = ridx[i]
        o_depth[starts[i]:starts[i] + n, 0] = (inv_scale * (base + k).astype(F32)).astype(F32)
        if first[i] and n > 0:
            o_bound[starts[i]] = True
    return dict(ridx=o_ridx, depth_samples=o_depth, boundary=o_bound)
