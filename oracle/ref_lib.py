"""Test infrastructure: ctypes access to oracle/_ref/libwisp_ref.so - the REFERENCE's own kernel bodies
(wisp/csrc/ops/hashgrid_interpolate_cuda.cu:19-339, hash_utils.cuh:17-112, uniform_sample_cuda.cu:18-59)
compiled for the host by oracle/build_ref.sh (the 3-D interpolation kernels carry a tap macro right after their floor() so that
`cells_3d` can read the scaled position and integer cell the reference code computed; `fma_cells` is NOT reference code: libm's fmaf
evaluation of this package's one-fma formula, kept here because the two are compared with each other).  Used to pin oracle/hashgrid.py and oracle/raymarch.uniform_sample
against the real reference arithmetic and to generate tests/golden/*.npz.  Absent => `available()` is False.  Also holds the
SDF tracer's find_depth_bound kernel body (render/find_depth_bound_cuda.cu:16-45) and the corner-query kernels (hashgrid_query_cuda.cu)."""
import ctypes
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libwisp_ref.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_PATH)
        _lib.ref_hash_index_3d.restype = ctypes.c_int32
        _lib.ref_hash_index_2d.restype = ctypes.c_int32
        _lib.ref_clamp.restype = ctypes.c_float
        _lib.ref_clamp.argtypes = [ctypes.c_float] * 3
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def hash_index_3d(x, y, z, res, T):
    return int(lib().ref_hash_index_3d(int(x), int(y), int(z), int(res), int(T)))


def hash_index_2d(x, y, res, T):
    return int(lib().ref_hash_index_2d(int(x), int(y), int(res), int(T)))


def clamp(x, a, b):
    return float(lib().ref_clamp(x, a, b))


def cells_3d(coords, res, codebook_size=2 ** 19):
    """(x f32 [N,3], pos i32 [N,3]): the clamped scaled position and its floor exactly as the reference's 3-D interpolation
    kernel computes them (hashgrid_interpolate_cuda.cu:40-43) - its own code, tapped."""
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    n = coords.shape[0]
    x = np.empty((n, 3), dtype=np.float32)
    pos = np.empty((n, 3), dtype=np.int32)
    lib().ref_cells_3d(ctypes.c_int64(n), ctypes.c_int32(int(res)), ctypes.c_int32(int(codebook_size)), _p(coords), _p(x), _p(pos))
    return x, pos


def fma_cells(coords, res):
    """Host evaluation of csrc/hashgrid.hip's one-fma position formula (libm fmaf), any shape of float32 coordinates."""
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    x = np.empty(coords.shape, dtype=np.float32)
    pos = np.empty(coords.shape, dtype=np.int32)
    lib().fma_cells(ctypes.c_int64(coords.size), ctypes.c_int32(int(res)), _p(coords), _p(x), _p(pos))
    return x, pos


def hashgrid_forward(coords, table, begin_idxes, resolutions, codebook_bitwidth):
    """hashgrid_interpolate_cuda (hashgrid_interpolate.cpp:46-69) with the reference kernels, float32 tables."""
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    table = np.ascontiguousarray(table, dtype=np.float32)
    begin = np.ascontiguousarray(begin_idxes, dtype=np.int64)
    n, dim = coords.shape
    L, F = len(resolutions), table.shape[1]
    feats = np.zeros((n, L * F), dtype=np.float32)
    for l, r in enumerate(resolutions):
        lib().ref_hashgrid_fwd_level(ctypes.c_int64(n), ctypes.c_int32(2 ** codebook_bitwidth), ctypes.c_int64(F),
                                     ctypes.c_int32(int(r)), ctypes.c_int32(l), ctypes.c_int32(L), ctypes.c_int(dim),
                                     _p(coords), _p(table), _p(begin), _p(feats))
    return feats


def hashgrid_backward(coords, grad_out, table, begin_idxes, resolutions, codebook_bitwidth):
    """hashgrid_interpolate_backward_cuda (hashgrid_interpolate.cpp:71-105), float32, sequential adds."""
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    table = np.ascontiguousarray(table, dtype=np.float32)
    grad_out = np.ascontiguousarray(grad_out, dtype=np.float32)
    begin = np.ascontiguousarray(begin_idxes, dtype=np.int64)
    n, dim = coords.shape
    L, F = len(resolutions), table.shape[1]
    grad = np.zeros_like(table)
    for l, r in enumerate(resolutions):
        lib().ref_hashgrid_bwd_level(ctypes.c_int64(n), ctypes.c_int32(2 ** codebook_bitwidth), ctypes.c_int64(F),
                                     ctypes.c_int32(int(r)), ctypes.c_int32(l), ctypes.c_int32(L), ctypes.c_int(dim),
                                     _p(coords), _p(table), _p(begin), _p(grad_out), _p(grad))
    return grad


def hashgrid_grad_coords(coords, grad_out, table, begin_idxes, resolutions, codebook_bitwidth):
    """grad_coords f32 [N, 3] of hashgrid_interpolate_backward_cuda(..., require_grad_coords=True) (hashgrid_interpolate.cpp:88-100):
    the reference's backward kernels with the flag set, one launch per level accumulating into the zero [N, 3] tensor."""
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    table = np.ascontiguousarray(table, dtype=np.float32)
    grad_out = np.ascontiguousarray(grad_out, dtype=np.float32)
    begin = np.ascontiguousarray(begin_idxes, dtype=np.int64)
    n, dim = coords.shape
    L, F = len(resolutions), table.shape[1]
    grad = np.zeros_like(table)
    gc = np.zeros((n, 3), dtype=np.float32)
    for l, r in enumerate(resolutions):
        lib().ref_hashgrid_bwd_level_coords(ctypes.c_int64(n), ctypes.c_int32(2 ** codebook_bitwidth), ctypes.c_int64(F),
                                            ctypes.c_int32(int(r)), ctypes.c_int32(l), ctypes.c_int32(L), ctypes.c_int(dim),
                                            _p(coords), _p(table), _p(begin), _p(grad_out), _p(grad), _p(gc))
    return gc


def hashgrid_query(coords, tables, resolutions, codebook_bitwidth, probe_bitwidth=0):
    """hashgrid_query_cuda (hashgrid_query.cpp:41-67) with the reference kernel, float32 tables (one per level):
    -> [N, 8, L, P, F]."""
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    n, L, F, P = coords.shape[0], len(resolutions), tables[0].shape[1], 2 ** probe_bitwidth
    feats = np.zeros((n, 8, L, P, F), dtype=np.float32)
    for l, r in enumerate(resolutions):
        t = np.ascontiguousarray(tables[l], dtype=np.float32)
        lib().ref_hashgrid_query_level(ctypes.c_int64(n), ctypes.c_int32(2 ** codebook_bitwidth), ctypes.c_int32(P),
                                       ctypes.c_int64(F), ctypes.c_int32(int(r)), ctypes.c_int32(l), ctypes.c_int32(L),
                                       _p(coords), _p(t), _p(feats))
    return feats


def hashgrid_query_backward(coords, grad_out, resolutions, codebook_bitwidth, feature_dim, probe_bitwidth=0):
    """hashgrid_query_backward_cuda (hashgrid_query.cpp:69-97), float32 (the path that adds probe p into row idx + p),
    sequential adds: -> list of [2^bw, F] tables."""
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    grad_out = np.ascontiguousarray(grad_out, dtype=np.float32)
    n, L, P = coords.shape[0], len(resolutions), 2 ** probe_bitwidth
    out = []
    for l, r in enumerate(resolutions):
        g = np.zeros((2 ** codebook_bitwidth, feature_dim), dtype=np.float32)
        lib().ref_hashgrid_query_bwd_level(ctypes.c_int64(n), ctypes.c_int32(2 ** codebook_bitwidth), ctypes.c_int32(P),
                                           ctypes.c_int64(feature_dim), ctypes.c_int32(int(r)), ctypes.c_int32(l),
                                           ctypes.c_int32(L), _p(coords), _p(grad_out), _p(g))
        out.append(g)
    return out


def find_depth_bound(query, curr_idxes, depth):
    """find_depth_bound_cuda (render/find_depth_bound.cpp:23-36 + the kernel): query f32 [P], curr_idxes i32 [P], depth f32 [M,2]
    -> i32 [P] (-1 where nothing is found)."""
    q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
    cur = np.ascontiguousarray(curr_idxes, dtype=np.int32)
    dep = np.ascontiguousarray(depth, dtype=np.float32)
    out = np.empty(q.shape[0], dtype=np.int32)
    lib().ref_find_depth_bound(ctypes.c_int64(q.shape[0]), ctypes.c_int64(dep.shape[0]), _p(q), _p(cur), _p(out), _p(dep))
    return out


def uniform_sample(scale, ridx, depth, insum):
    ridx = np.ascontiguousarray(ridx, dtype=np.int32)
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    insum = np.ascontiguousarray(insum, dtype=np.int32)
    V = ridx.shape[0]
    total = int(insum[-1]) if V else 0
    new_ridx = np.zeros(total, dtype=np.int64)
    ds = np.zeros((total, 1), dtype=np.float32)
    boundary = np.zeros(total, dtype=np.bool_)
    if V:
        lib().ref_uniform_sample(ctypes.c_int32(V), ctypes.c_float(scale), _p(ridx), _p(depth), _p(insum), _p(new_ridx),
                                 _p(ds), _p(boundary))
    return dict(ridx=new_ridx, depth_samples=ds, boundary=boundary)
