"""Oracle (test infrastructure): multi-resolution hash/dense grid interpolation, forward and backward.

Restates wisp/csrc/ops/hashgrid_interpolate_cuda.cu:19-339 (kernels), wisp/csrc/ops/hash_utils.cuh:17-112
(index functions), wisp/csrc/ops/hashgrid_interpolate.cpp:46-105 (per-LOD launch loop, output layout),
wisp/ops/grid.py:77-144 (autograd wrapper) and wisp/models/grids/utils.py:13-67 (table layout),
following the numerics contract in SURVEY.md Appendix B.  torch-CPU tensors; integer index math is
bit-exact, the float blend is float32 accumulated corner by corner.

Parity: PINNED to the reference's kernel bodies built for the host (oracle/_ref): forward bit-exact (3-D and 2-D), backward within fp32
add-order noise, corner query bit-exact (tests/golden/hashgrid_ref.npz, hashgrid_query_ref.npz); grid_interpolate equals
HashGrid.interpolate -> the reference's ops/grid.py -> those kernels bit for bit.
"""
import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


def _i32(v):
    """wrap a python int to int32 (the reference evaluates res*res*res in int32, hash_utils.cuh:27-29)."""
    return ((int(v) + 2 ** 31) % 2 ** 32) - 2 ** 31


def level_is_dense(res, codebook_size, coord_dim=3):
    """hash_utils.cuh:27-29 (3-D) / :75-76 (2-D): strict '<' on int32 products."""
    if coord_dim == 3:
        return res < codebook_size and _i32(res * res) < codebook_size and _i32(_i32(res * res) * res) < codebook_size
    return res < codebook_size and _i32(res * res) < codebook_size


def table_layout(resolutions, codebook_size, coord_dim=3):
    """models/grids/utils.py:48-60: num_feats[l] = min(T, res^coord_dim); begin_idxes = exclusive sum."""
    sizes = [min(codebook_size, int(r) ** coord_dim) for r in resolutions]
    begin = np.zeros(len(sizes) + 1, dtype=np.int64)
    begin[1:] = np.cumsum(sizes)
    return np.asarray(sizes, dtype=np.int64), begin


def corner_setup(coords, res, codebook_size):
    """Per level: scaled position, integer cell, trilinear/bilinear coefficients and corner indices.

    x_a = fl32(clamp(fl32(double(res) * (double(c_a)*0.5 + 0.5)), 0, fl32(res-1-1e-5)))   (.cu:40-42)
    pos = floor(x); f = x - pos; g = 1 - f; coeff_j = prod over axes (bit ? f : g), left to right  (.cu:43-56)
    index: dense x + y*res + z*res^2 (int32) or ((x*1) ^ (y*2654435761) ^ (z*805459861)) % T in uint32.
    Returns (coeffs float32 [N, 2^d], idx int64 [N, 2^d]).
    """
    N, dim = coords.shape
    x = (coords.double() * 0.5 + 0.5) * float(res)
    x = x.float()
    hi = float(np.float32(res - 1 - 1e-5))
    x = torch.clamp(x, min=0.0, max=hi)
    pos = torch.floor(x)
    f = x - pos
    g = 1.0 - f
    pos = pos.to(torch.int64)
    ncorner = 1 << dim
    coeffs = torch.empty(N, ncorner, dtype=torch.float32)
    idx = torch.empty(N, ncorner, dtype=torch.int64)
    dense = level_is_dense(int(res), int(codebook_size), dim)
    for j in range(ncorner):
        bits = [(j >> (dim - 1 - a)) & 1 for a in range(dim)]
        c = None
        for a in range(dim):
            term = f[:, a] if bits[a] else g[:, a]
            c = term if c is None else c * term
        coeffs[:, j] = c
        corner = [pos[:, a] + bits[a] for a in range(dim)]
        if dense:
            lin = corner[0] + corner[1] * res
            if dim == 3:
                lin = lin + corner[2] * (res * res)
            # int32 wrap of the reference; never triggers for res^3 < 2^31 but keeps the restatement honest
            idx[:, j] = ((lin + 2 ** 31) % 2 ** 32) - 2 ** 31
        else:
            h = (corner[0] * PRIMES[0]) & 0xFFFFFFFF
            for a in range(1, dim):
                h = h ^ ((corner[a] * PRIMES[a]) & 0xFFFFFFFF)
            idx[:, j] = h % int(codebook_size)
    return coeffs, idx


def hashgrid_forward(coords, table, begin_idxes, resolutions, codebook_bitwidth):
    """hashgrid_interpolate_cuda (hashgrid_interpolate.cpp:46-69): out[N, L*F], slot l*F..l*F+F-1 per
    level, dtype of `table`; accumulation in float32 in corner order j = 0..2^d-1 (.cu:68-78)."""
    coords = coords.float().reshape(-1, coords.shape[-1])
    N, dim = coords.shape
    L, F = len(resolutions), table.shape[1]
    T = 2 ** codebook_bitwidth
    out = torch.empty(N, L * F, dtype=table.dtype)
    tf = table.float()
    for l, res in enumerate(resolutions):
        res = int(res)
        coeffs, idx = corner_setup(coords, res, T)
        base = int(begin_idxes[l])
        acc = None
        for j in range(1 << dim):
            term = tf[base + idx[:, j]] * coeffs[:, j:j + 1]
            acc = term if acc is None else acc + term
        out[:, l * F:(l + 1) * F] = acc.to(table.dtype)
    return out


def hashgrid_backward(coords, grad_out, table_shape, begin_idxes, resolutions, codebook_bitwidth,
                      accum_dtype=torch.float32):
    """hashgrid_interpolate_backward_cuda (hashgrid_interpolate.cpp:71-105, .cu:106-161):
    grad_table[idx_j] += grad_out[:, l*F:(l+1)*F] * coeff_j.  The reference adds with atomics in the
    table dtype (order-dependent); the oracle accumulates in `accum_dtype` (float32 or float64)."""
    coords = coords.float().reshape(-1, coords.shape[-1])
    N, dim = coords.shape
    L, F = len(resolutions), table_shape[1]
    T = 2 ** codebook_bitwidth
    grad = torch.zeros(table_shape, dtype=accum_dtype)
    go = grad_out.reshape(N, L * F).to(accum_dtype)
    for l, res in enumerate(resolutions):
        coeffs, idx = corner_setup(coords, int(res), T)
        base = int(begin_idxes[l])
        g = go[:, l * F:(l + 1) * F]
        for j in range(1 << dim):
            grad.index_add_(0, base + idx[:, j], g * coeffs[:, j:j + 1].to(accum_dtype))
    return grad


def hashgrid_grad_coords(coords, grad_out, table, begin_idxes, resolutions, codebook_bitwidth):
    """grad_coords [N, 3] as hashgrid_interpolate_backward_cuda returns it with require_grad_coords (hashgrid_interpolate.cpp:
    88-100, kernel body hashgrid_interpolate_cuda.cu:163-196), restated as the reference computes it - NOT as calculus would:
      * every level reads the upstream gradient of the FIRST level's columns: grad_output[i*L*F + j] (.cu:165-166, the
        source's own "FIX IN MASTER lod_idx");
      * d/dy's last term is (x_.x * x_.z) * (corner 7 - corner 6); the trilinear formula has corner 5 there (.cu:185-186);
      * no factor res / 2 for d(cell position) / d(coordinate);
      * 2-D coordinates: the 2-D kernel takes the flag and writes nothing - zeros [N, 3].
    float32, levels in order, features in order, the four products of a component summed left to right."""
    coords = coords.float().reshape(-1, coords.shape[-1])
    N, dim = coords.shape
    L, F = len(resolutions), table.shape[1]
    out = torch.zeros(N, 3, dtype=torch.float32)
    if dim != 3:
        return out
    T = 2 ** codebook_bitwidth
    tf = table.float()
    go = grad_out.reshape(N, L * F).float()
    for l, res in enumerate(resolutions):
        res = int(res)
        _, idx = corner_setup(coords, res, T)
        x = ((coords.double() * 0.5 + 0.5) * float(res)).float()
        x = torch.clamp(x, min=0.0, max=float(np.float32(res - 1 - 1e-5)))
        f = x - torch.floor(x)                                     # x_
        g = 1.0 - f                                                # _x
        base = int(begin_idxes[l])
        for j in range(F):
            v = [tf[base + idx[:, k], j] for k in range(8)]
            w = go[:, j]                                           # (level 0's columns, whatever l)
            out[:, 0] += w * ((g[:, 1] * g[:, 2]) * (v[4] - v[0]) + (g[:, 1] * f[:, 2]) * (v[5] - v[1])
                              + (f[:, 1] * g[:, 2]) * (v[6] - v[2]) + (f[:, 1] * f[:, 2]) * (v[7] - v[3]))
            out[:, 1] += w * ((g[:, 0] * g[:, 2]) * (v[2] - v[0]) + (g[:, 0] * f[:, 2]) * (v[3] - v[1])
                              + (f[:, 0] * g[:, 2]) * (v[6] - v[4]) + (f[:, 0] * f[:, 2]) * (v[7] - v[6]))
            out[:, 2] += w * ((g[:, 0] * g[:, 1]) * (v[1] - v[0]) + (g[:, 0] * f[:, 1]) * (v[3] - v[2])
                              + (f[:, 0] * g[:, 1]) * (v[5] - v[4]) + (f[:, 0] * f[:, 1]) * (v[7] - v[6]))
    return out


class HashGridInterpolate(torch.autograd.Function):
    """wisp/ops/grid.py:77-126 on the CPU oracle (no autocast branch; table dtype drives the output)."""

    @staticmethod
    def forward(ctx, coords, resolutions, codebook_bitwidth, lod_idx, codebook, codebook_first_idx):
        if codebook.shape[-1] % 2 == 1:
            raise Exception("The codebook feature dimension needs to be a multiple of 2.")
        assert coords.shape[-1] in (2, 3)
        ctx.save_for_backward(coords, codebook_first_idx)
        ctx.meta = (list(int(r) for r in resolutions), codebook_bitwidth, tuple(codebook.shape), codebook.dtype)
        return hashgrid_forward(coords.detach(), codebook.detach(), codebook_first_idx, ctx.meta[0], codebook_bitwidth)

    @staticmethod
    def backward(ctx, grad_output):
        coords, first_idx = ctx.saved_tensors
        res, bw, shape, dtype = ctx.meta
        g = hashgrid_backward(coords, grad_output, shape, first_idx, res, bw)
        return None, None, None, None, g.to(dtype), None


def hashgrid(coords, resolutions, codebook_bitwidth, lod_idx, table, begin_idxes):
    """wisp/ops/grid.py:128-144."""
    return HashGridInterpolate.apply(coords.contiguous(), resolutions, codebook_bitwidth, lod_idx, table, begin_idxes)


def grid_interpolate(coords, lod_idx, multiscale_type, feature_dim, resolutions, codebook_bitwidth, table, begin_idxes):
    """HashGrid.interpolate (models/grids/hash_grid.py:205-233) including the 'cat' quirk that zeroes
    feats[..., lod_idx*F:] (:226-229)."""
    out_shape = coords.shape[:-1]
    flat = coords.reshape(-1, coords.shape[-1])
    feats = hashgrid(flat, resolutions, codebook_bitwidth, lod_idx, table, begin_idxes)
    if multiscale_type == 'cat':
        feats = feats.reshape(*out_shape, feats.shape[-1])
        keep = torch.ones(feats.shape[-1], dtype=feats.dtype)
        keep[lod_idx * feature_dim:] = 0
        return feats * keep
    elif multiscale_type == 'sum':
        L = len(resolutions)
        return feats.reshape(*out_shape, L, feats.shape[-1] // L).sum(-2)
    raise NotImplementedError


def hashgrid_query(coords, tables, resolutions, codebook_bitwidth, probe_bitwidth=0):
    """wisp._C.ops.hashgrid_query_cuda (wisp/csrc/ops/hashgrid_query_cuda.cu:19-66, hashgrid_query.cpp:41-67): the eight corner
    rows of every level, un-blended.  One table [2^bw, F] per level; corner k = dx<<2 | dy<<1 | dz; index = hash_index_3d(corner,
    res, 2^bw - P) with P = 2^probe_bitwidth; the row is written to ALL P probe slots (the kernel reads row idx for every p).
    -> [N, 8, L, P, F] in the tables' dtype.  3-D coordinates only (the kernel reads coords[i*3 ..])."""
    N = coords.shape[0]
    L, F, P = len(resolutions), tables[0].shape[1], 2 ** probe_bitwidth
    mod = 2 ** codebook_bitwidth - P
    out = torch.zeros(N, 8, L, P, F, dtype=tables[0].dtype)
    for l, r in enumerate(resolutions):
        _, idx = corner_setup(coords, int(r), mod)
        rows = tables[l][idx.reshape(-1)].reshape(N, 8, 1, F)
        out[:, :, l] = rows.expand(N, 8, P, F)
    return out


def hashgrid_query_backward(coords, grad_out, resolutions, codebook_bitwidth, feature_dim, probe_bitwidth=0, half_path=False,
                            acc_dtype=torch.float64):
    """hashgrid_query_backward_cuda (hashgrid_query_cuda.cu:100-171): scatter-add of grad_out [N, 8, L, P, F] into one table
    [2^bw, F] per level.  fp32 path (:157-166): probe p goes to row idx + p; half path (:141-155): every probe to row idx."""
    N = coords.shape[0]
    L, P = len(resolutions), 2 ** probe_bitwidth
    mod = 2 ** codebook_bitwidth - P
    g = grad_out.reshape(N, 8, L, P, feature_dim).to(acc_dtype)
    out = []
    for l, r in enumerate(resolutions):
        _, idx = corner_setup(coords, int(r), mod)
        t = torch.zeros(2 ** codebook_bitwidth, feature_dim, dtype=acc_dtype)
        for p in range(P):
            rows = (idx if half_path else idx + p).reshape(-1)
            t.index_add_(0, rows, g[:, :, l, p].reshape(-1, feature_dim))
        out.append(t)
    return out

