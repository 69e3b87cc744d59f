"""ctypes binding of libwisp_hip.so - the C ABI declared in include/wisp_hip.h.

Plays the role of the reference's pybind module ``wisp._C`` (wisp/csrc/bindings.cpp:21-35) plus the
Kaolin-Core leaves wisp calls.  There is NO fallback: if the shared library is missing, importing this
module raises, and every op refuses tensors that are not on the GPU.
The functions below are thin: they allocate outputs from torch (so the caching allocator and the
current stream stay PyTorch's), pass raw device pointers + the current HIP stream, and check the status.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# WISP_HIP_LIB points at another build of the same C ABI (an installed copy, or an A/B variant while tuning kernels)
LIB_PATH = os.environ.get("WISP_HIP_LIB") or os.path.normpath(os.path.join(_HERE, "..", "csrc", "libwisp_hip.so"))

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"wisp HIP extension not built: {LIB_PATH} is missing. Build it with "
        f"`make -C {os.path.dirname(LIB_PATH)}` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "There is no CPU fallback for the hot path.")

lib = ctypes.CDLL(LIB_PATH)

F32, F16, BF16 = 0, 1, 2
_DTYPE_CODE = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

c_vp, c_i64, c_i32, c_f32, c_u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_uint64

# name -> argtypes (restype int unless listed in _RESTYPES); mirrors include/wisp_hip.h one to one
SIGNATURES = {
    "wisp_hashgrid_interpolate_fwd": [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp],
    "wisp_hashgrid_interpolate_bwd": [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp],
    "wisp_hashgrid_interpolate_bwd_adamw": [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp,
                                            c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, c_f32, c_vp, c_vp],
    "wisp_hashgrid_grad_coords": [c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp],
    "wisp_hashgrid_bwd_workspace_bytes": [c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp],
    "wisp_hashgrid_bwd_slot_stats": [c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp],
    "wisp_hashgrid_cells": [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp],
    "wisp_hashgrid_query_fwd": [c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp],
    "wisp_hashgrid_query_bwd": [c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp],
    "wisp_spc_query": [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp],
    "wisp_spc_query_chain": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp],
    "wisp_spc_build_bitfield": [c_vp, c_i64, c_i32, c_vp, c_vp],
    "wisp_spc_raytrace_count": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp],
    "wisp_spc_raytrace_emit": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp],
    "wisp_spc_trilinear_coeffs": [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp],
    "wisp_spc_trilinear_fwd": [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp],
    "wisp_spc_trilinear_bwd": [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp],
    "wisp_spc_bwd_workspace_bytes": [c_i64, c_i32, c_i64],
    "wisp_spc_trilinear_multi_fwd": [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp],
    "wisp_spc_trilinear_multi_bwd": [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp],
    "wisp_triplane_fwd": [c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp],
    "wisp_triplane_bwd": [c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp],
    "wisp_codebook_trilinear_fwd": [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp],
    "wisp_codebook_decode_rows": [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp],
    "wisp_codebook_trilinear_bwd": [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp],
    "wisp_codebook_trilinear_multi_bwd": [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp,
                                          c_vp, c_i64, c_vp],
    "wisp_mark_pack_boundaries_i64": [c_vp, c_i64, c_vp, c_vp],
    "wisp_mark_pack_boundaries_i32": [c_vp, c_i64, c_vp, c_vp],
    "wisp_scan_workspace_bytes": [c_i64],
    "wisp_exclusive_scan_i32": [c_vp, c_i64, c_vp, c_vp, c_vp],
    "wisp_inclusive_scan_i32": [c_vp, c_i64, c_vp, c_vp, c_vp],
    "wisp_boundary_tile_counts": [c_vp, c_i64, c_vp, c_vp],
    "wisp_boundary_pack_starts": [c_vp, c_i64, c_vp, c_vp, c_vp],
    "wisp_raymarch_ray_count": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_i32, c_i32, c_vp, c_u64, c_vp, c_i32, c_vp, c_vp, c_vp],
    "wisp_raymarch_ray_emit": [c_vp, c_vp, c_i64, c_f32, c_f32, c_i32, c_vp, c_u64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wisp_raymarch_voxel_emit": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_u64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wisp_raymarch_uniform_count": [c_vp, c_i64, c_f32, c_vp, c_vp],
    "wisp_spc_mask_from_points": [c_vp, c_i64, c_i32, c_vp, c_vp],
    "wisp_spc_dense_bytes": [c_vp, c_i32, c_vp],
    "wisp_spc_points_from_index": [c_vp, c_i64, c_i32, c_vp, c_vp],
    "wisp_sdf_trace_step_fused": [c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                  c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                  c_vp, c_i32, c_f32, c_vp, c_vp],
    "wisp_sdf_train_scratch_bytes": [c_i64, c_i32, c_i32, c_i32],
    "wisp_sdf_train_step": [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                            c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp],
    "wisp_grid_interpolate_fwd": [c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp],
    "wisp_grid_interpolate_bwd": [c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp],
    "wisp_small_decoder_fwd": [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wisp_small_decoder_bwd": [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wisp_uniform_sample": [c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp],
    "wisp_raymarch_uniform_emit": [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wisp_packed_sum_reduce": [c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp],
    "wisp_packed_cumsum": [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp],
    "wisp_find_depth_bound": [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp],
    "wisp_sphere_trace_step": [c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wisp_optim_step_groups": [c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_i64, c_f32, c_i32, c_vp],
    "wisp_adamw_step_groups": [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_i64, c_f32, c_i32, c_vp],
    "wisp_gather_rows": [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp],
    "wisp_composite_loss": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "wisp_rgb_loss": [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp],
    "wisp_generate_rays": [c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wisp_composite_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wisp_composite_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp],
    "wisp_nerf_mlp_param_count": [c_i32, c_i32, c_i32],
    "wisp_nerf_mlp_fwd": [c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp],
    "wisp_nerf_mlp_bwd": [c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "wisp_nerf_mlp_dir_code": [c_vp, c_i64, c_i32, c_vp, c_vp],
    "wisp_nerf_mlp_fwd_rays": [c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp],
    "wisp_nerf_mlp_bwd_rays": [c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "wisp_nerf_mlp_bwd_workspace_bytes": [c_i64, c_i32],
    "wisp_nerf_mlp_workspace_floats": [],
    "wisp_adamw_step": [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, c_f32, c_i32, c_vp, c_vp],
    "wisp_host_reader_create": [],
    "wisp_host_reader_issue": [c_vp, c_vp, c_vp],
    "wisp_host_reader_wait": [c_vp, c_vp],
    "wisp_host_reader_destroy": [c_vp],
    "wisp_nerf_step_config_bytes": [],
    "wisp_nerf_step_workspace_bytes": [c_vp],
    "wisp_nerf_step_create": [c_vp, c_vp, c_i64],
    "wisp_nerf_step_reconfigure": [c_vp, c_vp],
    "wisp_nerf_step_destroy": [c_vp],
    "wisp_nerf_step_count": [c_vp, c_i32, c_vp, c_vp, c_i64, c_u64, c_vp],
    "wisp_nerf_step_run": [c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_u64, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp],
    "wisp_nerf_step_read_timing": [c_vp, c_i32, c_vp, c_vp, c_vp],
    "wisp_last_error": [],
    "wisp_abi_version": [],
}
_RESTYPES = {"wisp_nerf_mlp_bwd_workspace_bytes": c_i64, "wisp_spc_bwd_workspace_bytes": c_i64, "wisp_sdf_train_scratch_bytes": c_i64, "wisp_hashgrid_bwd_workspace_bytes": c_i64, "wisp_scan_workspace_bytes": c_i64, "wisp_nerf_mlp_param_count": c_i64, "wisp_nerf_mlp_workspace_floats": c_i64,
             "wisp_last_error": ctypes.c_char_p, "wisp_host_reader_create": c_vp, "wisp_host_reader_destroy": None,
             "wisp_nerf_step_config_bytes": c_i64, "wisp_nerf_step_workspace_bytes": c_i64, "wisp_nerf_step_create": c_vp,
             "wisp_nerf_step_destroy": None}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here == header/library mismatch: fail loudly
    _fn.argtypes = _args
    _fn.restype = _RESTYPES.get(_name, c_i32)

ABI_VERSION = 4                        # include/wisp_hip.h: wisp_abi_version(); the signatures above are this version's
if lib.wisp_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} implements ABI version {lib.wisp_abi_version()}, this binding is written for version "
                      f"{ABI_VERSION}: rebuild the library (make -C {os.path.dirname(LIB_PATH)})")


WISP_ERR_CAPACITY = -4                 # include/wisp_hip.h


class NerfStepConfig(ctypes.Structure):
    """include/wisp_hip.h: wisp_nerf_step_config, field for field (sizeof is checked against the library's at import)."""
    _fields_ = ([("struct_bytes", c_i64)]
                + [(n, c_vp) for n in ("occ_bits", "octree", "exsum", "coarse_bits", "table_lookup", "first_idx", "first_idx_host",
                                       "resolutions", "table_param", "table_grad", "table_exp_avg", "table_exp_avg_sq", "table_shadow",
                                       "dec_params", "dec_grad", "flat_param", "flat_grad", "flat_exp_avg", "flat_exp_avg_sq",
                                       "grid_shadow")]
                + [(n, c_i64) for n in ("decoder_begin", "decoder_len", "grid_begin", "grid_len", "rest_begin", "rest_len", "table_offset",
                                        "max_rays", "max_samples")]
                + [(n, c_i32) for n in ("level", "coarse_level", "num_samples", "loss_kind", "dtype_table", "num_lods", "feature_dim",
                                        "bitwidth", "zero_from_col", "in_dim", "hidden", "view_freqs")]
                + [("near", c_f32), ("range", c_f32), ("bg", c_f32 * 3), ("reserved", c_f32)])


class NerfStepHyper(ctypes.Structure):
    """include/wisp_hip.h: wisp_nerf_step_hyper."""
    _fields_ = ([("struct_bytes", c_i64), ("step", c_i64)]
                + [(n, c_f32) for n in ("lr_decoder", "lr_grid", "lr_rest", "weight_decay", "beta1", "beta2", "eps", "grad_scale")]
                + [("optimizer", c_i32), ("reserved", c_i32)])


if ctypes.sizeof(NerfStepConfig) != lib.wisp_nerf_step_config_bytes():
    raise ImportError(f"wisp_nerf_step_config: this binding lays it out in {ctypes.sizeof(NerfStepConfig)} bytes, {LIB_PATH} in "
                      f"{lib.wisp_nerf_step_config_bytes()}: rebuild the library")


# Live HIP-event timing of selected kernels (bench.py sets TIMING = {} around its timed region).  Events are
# recorded on the stream the kernel is launched on (torch's current stream).
TIMING = None
# TIMING_ALL = {} additionally brackets EVERY C-ABI entry point that launches work with events, keyed by symbol name
# (the benches of the other configs use it to find the dominant kernel of a step).
TIMING_ALL = None
_cdll = lib


class _LibProxy:
    """`lib.<symbol>`: the ctypes function itself, or - for entry points that take a stream - a wrapper that records a pair
    of events around the call while TIMING_ALL is set.  One dict lookup of overhead otherwise."""

    def __getattr__(self, name):
        fn = getattr(_cdll, name)
        sig = SIGNATURES.get(name)
        if not sig or sig[-1] is not c_vp or name.endswith("_bytes") or name.endswith("_count") and "raymarch" not in name and "raytrace" not in name:
            setattr(self, name, fn)
            return fn

        def call(*args):
            sink = TIMING_ALL
            if sink is None:
                return fn(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            sink.setdefault(name, []).append((e0, e1))
            return rc
        setattr(self, name, call)
        return call


lib = _LibProxy()


class _timed:
    def __init__(self, name, units):
        self.name, self.units = name, units

    def __enter__(self):
        if TIMING is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if TIMING is not None:
            self.e1.record()
            TIMING.setdefault(self.name, []).append((self.e0, self.e1, self.units))
        return False


def last_error():
    return lib.wisp_last_error().decode()


def _check(rc, what):
    if rc != 0:
        # a failed call may have stopped between a scatter pass and the pass that restores its scratch to zero: buffers that later
        # calls rely on being all zero are dropped from their caches (ones baked into a captured graph stay alive, see _Scratch)
        for cache in _ZEROED_SCRATCH:
            cache.drop_all()
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


class _Scratch:
    """Per-(device, stream) scratch buffers that only ever grow.  A buffer that was handed out while its stream was being captured
    into a HIP graph has its address baked into that graph: it is kept alive for the life of the process and is never the one
    that gets replaced or freed - a later call that needs more gets a NEW buffer for eager use while the captured one stays
    where the graph expects it (and, for the zero-on-return kind, stays zero because only the graph writes to it)."""

    def __init__(self, zeroed):
        self.zeroed = zeroed
        self.live = {}
        self.captured = []                 # strong references: tensors some captured graph points into

    def get(self, device, need):
        key = (device.index if device.index is not None else torch.cuda.current_device(), _stream().value)
        capturing = bool(torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
        buf = self.live.get(key)
        if buf is None or buf.numel() < need:
            make = torch.zeros if self.zeroed else torch.empty
            buf = self.live[key] = make(need + need // 4, dtype=torch.uint8, device=device)
        if capturing and not any(buf is c for c in self.captured):
            self.captured.append(buf)
        return buf

    def drop_all(self):
        self.live.clear()


_ZEROED_SCRATCH = []


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream          # ~0.3 us; torch.cuda.current_stream() costs ~12 us per call
except AttributeError:                                         # pragma: no cover
    _raw_stream = None


_raw_device = getattr(torch._C, "_cuda_getDevice", None)       # (torch.cuda.current_device() walks through _lazy_init: 1 us per call)


def _stream():
    """the HIP stream torch is currently launching on (the kernels must be ordered with torch's own work)."""
    if _raw_stream is not None:
        return c_vp(_raw_stream(_raw_device() if _raw_device is not None else torch.cuda.current_device()))
    return c_vp(torch.cuda.current_stream().cuda_stream)


def _p(t):
    """device pointer of a tensor (or NULL)."""
    return c_vp(0) if t is None else c_vp(t.data_ptr())


def _need(t, dtype=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"wisp HIP op: `{name}` must be a GPU tensor (the hot path has no CPU fallback); got "
                           f"{type(t).__name__} on {getattr(t, 'device', None)}")
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


def _host_i32(values):
    arr = np.ascontiguousarray(np.asarray(values, dtype=np.int32).reshape(-1))
    return arr, arr.ctypes.data_as(c_vp)


_res_cache = {}


def _host_res(resolutions):
    """(tuple, numpy int32 array, pointer) of a level-resolution list - built once per distinct list: the per-step callers hand
    in the same sixteen integers every time, and numpy + ctypes conversions were 10 us of host time per step."""
    key = resolutions if type(resolutions) is tuple else tuple(int(r) for r in resolutions)
    hit = _res_cache.get(key)
    if hit is None:
        arr, ptr = _host_i32(key)
        hit = _res_cache[key] = (key, arr, ptr)
    return hit


def _host_f32(values):
    arr = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
    return arr, arr.ctypes.data_as(c_vp)


# ------------------------------------------------------------------------------------------------ hash grid
_first_idx_ok = {}


def _check_first_idx(first_idx, num_lods, rows):
    """codebook_first_idx is [L+1] (hashgrid_interpolate.h:18-23): the kernels read entry L as the end of the table, so a
    short or inconsistent vector would let them write past it.  Checked once per (tensor, version): one D2H read."""
    if first_idx.numel() < num_lods + 1:
        raise RuntimeError(f"codebook_first_idx must hold num_lods + 1 = {num_lods + 1} offsets, got {first_idx.numel()}")
    key = (first_idx.data_ptr(), first_idx._version, num_lods, rows)
    if key not in _first_idx_ok:
        if len(_first_idx_ok) > 64:
            _first_idx_ok.clear()
        host = [int(v) for v in first_idx[:num_lods + 1].tolist()]
        if host[num_lods] > rows:
            raise RuntimeError(f"codebook_first_idx[{num_lods}] = {host[num_lods]} exceeds the {rows} rows of the table")
        _first_idx_ok[key] = host
    return _first_idx_ok[key]


def hashgrid_interpolate(coords, codebook, first_idx, resolutions, codebook_bitwidth, zero_from_col=None):
    """feats[N, L*F] - the reference's wisp._C.ops.hashgrid_interpolate_cuda (hashgrid_interpolate.cpp:46-69)."""
    coords = _need(coords, torch.float32, "coords")
    codebook = _need(codebook, None, "codebook")
    first_idx = _need(first_idx, torch.int64, "codebook_first_idx")
    n, dim = coords.shape
    L, F = len(resolutions), codebook.shape[1]
    _check_first_idx(first_idx, L, codebook.shape[0])
    if zero_from_col is None:
        zero_from_col = L * F
    _, res_arr, res_ptr = _host_res(resolutions)
    feats = torch.empty(n, L * F, dtype=codebook.dtype, device=coords.device)
    with _timed("hashgrid_fwd", n):
        _check(lib.wisp_hashgrid_interpolate_fwd(_p(coords), n, dim, _p(codebook), _DTYPE_CODE[codebook.dtype], F,
                                                 _p(first_idx), res_ptr, L, codebook_bitwidth, zero_from_col, _p(feats),
                                                 _stream()), "hashgrid_interpolate_fwd")
    return feats


def hashgrid_interpolate_backward(coords, grad_feats, codebook_shape, first_idx, resolutions, codebook_bitwidth,
                                  zero_from_col=None, out=None, adamw=None):
    """fp32 grad_codebook - wisp._C.ops.hashgrid_interpolate_backward_cuda (hashgrid_interpolate.cpp:71-105).
    `out` (fp32, same shape) is accumulated into when given.
    adamw (needs `out`): dict(param, exp_avg, exp_avg_sq [fp32, the table's shape], shadow [bf16 or None], lr, beta1, beta2, eps,
    weight_decay, step, grad_scale) - the table's AdamW step folded into the backward (wisp_hashgrid_interpolate_bwd_adamw);
    returns (grad, covered) then, covered[l] = leading rows of level l that were updated inside the launch (their gradient is
    not in `out`); all other rows are the caller's to update."""
    coords = _need(coords, torch.float32, "coords")
    grad_feats = _need(grad_feats, None, "grad_feats")
    first_idx = _need(first_idx, torch.int64, "codebook_first_idx")
    n, dim = coords.shape
    L, F = len(resolutions), codebook_shape[1]
    first_host = _check_first_idx(first_idx, L, codebook_shape[0])
    if zero_from_col is None:
        zero_from_col = L * F
    res_key, res_arr, res_ptr = _host_res(resolutions)
    grad = out if out is not None else torch.zeros(tuple(codebook_shape), dtype=torch.float32, device=coords.device)
    assert grad.dtype == torch.float32 and grad.is_contiguous()
    dt = _DTYPE_CODE[grad_feats.dtype]
    # scratch for the binned reduction; its record slots are sized from what earlier launches of this shape really filled
    fit = _slot_fit(coords.device, dim, dt, F, res_key, codebook_bitwidth, zero_from_col, n >= HASHGRID_EMIT_WIDE_MIN) if n >= 4096 else None
    scale_arr, scale_ptr = fit.scales(n) if fit is not None else (None, None)
    ws_bytes = int(lib.wisp_hashgrid_bwd_workspace_bytes(n, dim, dt, F, res_ptr, L, codebook_bitwidth, scale_ptr))
    ws = _bwd_workspace(coords.device, ws_bytes) if 0 < ws_bytes <= HASHGRID_BWD_WORKSPACE_LIMIT else None
    covered = None
    with _timed("hashgrid_bwd", n):
        if adamw is None:
            _check(lib.wisp_hashgrid_interpolate_bwd(_p(coords), n, dim, _p(grad_feats), dt, F,
                                                     _p(first_idx), res_ptr, L, codebook_bitwidth, zero_from_col, _p(grad),
                                                     _p(ws), ws.numel() if ws is not None else 0, scale_ptr, _stream()),
                   "hashgrid_interpolate_bwd")
        else:
            assert out is not None, "the fused update leaves the uncovered rows' gradient in `out`"
            prm, m1, m2, sh = adamw["param"], adamw["exp_avg"], adamw["exp_avg_sq"], adamw.get("shadow")
            for t in (prm, m1, m2):
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == grad.numel()
            assert sh is None or (sh.dtype == torch.bfloat16 and sh.is_contiguous() and sh.numel() == grad.numel())
            cov = (ctypes.c_int64 * L)()
            _check(lib.wisp_hashgrid_interpolate_bwd_adamw(_p(coords), n, dim, _p(grad_feats), dt, F,
                                                           _p(first_idx), res_ptr, L, codebook_bitwidth, zero_from_col, _p(grad),
                                                           _p(ws), ws.numel() if ws is not None else 0, scale_ptr,
                                                           _p(prm), _p(m1), _p(m2), _p(sh), float(adamw["lr"]), float(adamw["beta1"]),
                                                           float(adamw["beta2"]), float(adamw["eps"]), float(adamw["weight_decay"]),
                                                           int(adamw["step"]), float(adamw.get("grad_scale", 1.0)),
                                                           ctypes.cast(cov, ctypes.c_void_p), _stream()),
                   "hashgrid_interpolate_bwd_adamw")
            # (levels stepped whole by the launch's tail workgroups report 2^62: rows as THIS table's first_idx spaces them)
            covered = [min(int(c), first_host[l + 1] - first_host[l]) for l, c in enumerate(cov)]
    if fit is not None and ws is not None:
        fit.after_launch(n, res_ptr, scale_arr, scale_ptr, ws, ws_bytes)
    return grad if adamw is None else (grad, covered)


def hashgrid_grad_coords(coords, grad_feats, codebook, first_idx, resolutions, codebook_bitwidth):
    """f32 [n, 3]: what hashgrid_interpolate_backward_cuda(..., require_grad_coords=True) returns as grad_coords
    (hashgrid_interpolate.cpp:88-90 + hashgrid_interpolate_cuda.cu:163-196, reproduced term by term - see include/wisp_hip.h).
    grad_feats [n, L*F] and codebook [rows, F] must share a dtype (the reference dispatches on one scalar type, .cu:413)."""
    coords = _need(coords, torch.float32, "coords")
    codebook = _need(codebook, None, "codebook")
    if grad_feats.dtype != codebook.dtype:
        grad_feats = grad_feats.to(codebook.dtype)
    grad_feats = _need(grad_feats, None, "grad_feats")
    first_idx = _need(first_idx, torch.int64, "codebook_first_idx")
    n, dim = coords.shape
    L, F = len(resolutions), codebook.shape[1]
    _check_first_idx(first_idx, L, codebook.shape[0])
    if grad_feats.shape != (n, L * F):
        raise ValueError(f"grad_feats must be [{n}, {L * F}], got {tuple(grad_feats.shape)}")
    res_arr, res_ptr = _host_i32(resolutions)
    out = torch.empty(n, 3, dtype=torch.float32, device=coords.device)
    _check(lib.wisp_hashgrid_grad_coords(_p(coords), n, dim, _p(grad_feats), _p(codebook), _DTYPE_CODE[codebook.dtype], F,
                                         _p(first_idx), res_ptr, L, int(codebook_bitwidth), _p(out), _stream()), "hashgrid_grad_coords")
    return out


HASHGRID_BWD_WORKSPACE_LIMIT = 24 << 30          # bytes; 288 GB of HBM makes a multi-GB scratch a fair trade
_bwd_ws = {}
_slot_fits = {}
# WISP_HG_SLOT_FIT=0: always the unscaled (no-merge expectation) slots
SLOT_FIT_ENABLED = os.environ.get("WISP_HG_SLOT_FIT", "1") != "0"


class _SlotFit:
    """Per-level record-slot sizing of the binned hash-grid backward, learned from the launches themselves.
    Every CHECK_EVERY-th launch of a shape is followed by wisp_hashgrid_bwd_slot_stats (one tiny kernel + a 64-byte copy to pinned
    memory, never waited for); a later launch picks the result up once the copy has landed: a level whose fullest slot stayed
    below its capacity gets slots of HEADROOM x that fill (as a fraction of the unscaled capacity, which follows the sample
    count), a level that filled a slot completely - it overflowed into the atomic path - gets GROW x its previous size.
    HEADROOM is 1.2 (WISP_HG_SLOT_HEADROOM): the fullest slot of a level moves by 1-5 % from launch to launch at a fixed batch size
    (profiles/r05_ab_scratch_geometry.txt); it was 1.35 until round 5."""
    CHECK_EVERY, HEADROOM, GROW, FLOOR, ADOPT_AFTER = 32, float(os.environ.get("WISP_HG_SLOT_HEADROOM", "1.2")), 1.6, 0.02, 8

    def __init__(self, device, dim, dt, F, res, bitwidth, zero_from_col):
        self.key = (dim, dt, F, res, bitwidth, zero_from_col)
        self.L = len(res)
        self.scale = [1.0] * self.L
        self.calls = 0
        self.pending = None
        self.last = None            # what the last evaluated check saw (tests, bench)
        self.dev_fill = torch.zeros(2 * self.L, dtype=torch.int32, device=device)      # [fullest slot x L | records written x L]
        self.host_fill = torch.zeros(2 * self.L, dtype=torch.int32).pin_memory()
        self._scale_c = None

    def scales(self, n):
        """-> (ctypes float array, its pointer); rebuilt only when a check changed the scales"""
        self._collect()
        if self._scale_c is None or self._scale_c[2] != self.scale:
            arr = (ctypes.c_float * self.L)(*self.scale)
            self._scale_c = (arr, ctypes.cast(arr, ctypes.c_void_p), list(self.scale))
        return self._scale_c[0], self._scale_c[1]

    def _collect(self):
        p = self.pending
        if p is None:
            return
        # A check is adopted at a FIXED launch number (ADOPT_AFTER launches after it was taken), not "whenever the copy happens to
        # have landed": which launch first runs with the new slot sizes then does not depend on host timing, and neither does
        # which records overflow a slot into the float-atomic path - same inputs, same bits, run after run.  The copy is long
        # done by then (the wait below returns at once); WISP_HG_SLOT_FIT=0 switches the whole mechanism off.
        if self.calls - p["at"] < self.ADOPT_AFTER:
            return
        p["event"].synchronize()
        self.pending = None
        both = self.host_fill.tolist()
        fill, written = both[:self.L], both[self.L:]
        for l in range(self.L):
            cap, base = p["cap"][l], p["base"][l]
            if base <= 0:
                continue
            if fill[l] >= cap:
                self.scale[l] = min(1.0, max(self.scale[l], cap / base) * self.GROW)
            else:
                self.scale[l] = min(1.0, max(self.FLOOR, self.HEADROOM * fill[l] / base))
        self.last = dict(fill=fill, records=written, cap=list(p["cap"]), base=list(p["base"]), scale=list(self.scale),
                         workspace_bytes=p["ws_bytes"])

    def after_launch(self, n, res_ptr, scale_arr, scale_ptr, ws, ws_bytes):
        self.calls += 1
        if self.pending is not None or (self.calls - 1) % self.CHECK_EVERY:
            return
        dim, dt, F, res, bitwidth, zero_from_col = self.key
        cap = (ctypes.c_int32 * self.L)()
        base = (ctypes.c_int32 * self.L)()
        rc = lib.wisp_hashgrid_bwd_slot_stats(n, dim, dt, F, res_ptr, self.L, bitwidth, zero_from_col, scale_ptr, _p(ws), ws.numel(),
                                              _p(self.dev_fill), ctypes.cast(cap, ctypes.c_void_p), ctypes.cast(base, ctypes.c_void_p),
                                              _stream())
        if rc < 0:
            _check(rc, "hashgrid_bwd_slot_stats")
        if rc != 0:
            return                  # the launch was not binned: nothing to learn
        self.host_fill.copy_(self.dev_fill, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = dict(event=ev, cap=list(cap), base=list(base), ws_bytes=int(ws_bytes), at=self.calls)


# csrc/hashgrid.hip EQ_WIDE_MIN: launches of at least this many samples use the 1024-thread emitter (2048-sample pieces).  The
# fullest-slot-to-mean ratio differs between the two emitter widths (1.6-2.1 x wide, 1.9-2.6 x narrow), so slot scales learned in
# one width must not be applied to the other: the width is part of the fit's key.  tests/test_abi_and_host.py holds
# this constant to the #define.
HASHGRID_EMIT_WIDE_MIN = 1 << 20


def _slot_fit(device, dim, dt, F, res, bitwidth, zero_from_col, wide=False):
    if not SLOT_FIT_ENABLED:
        return None
    key = (device, _stream().value, dim, dt, F, res, bitwidth, zero_from_col, bool(wide))
    fit = _slot_fits.get(key)
    if fit is None:
        fit = _slot_fits[key] = _SlotFit(device, dim, dt, F, res, bitwidth, zero_from_col)
    return fit


def hashgrid_cells(coords, resolution, codebook_bitwidth, with_corners=True):
    """(cell i32 [N,d], frac f32 [N,d], corner rows i32 [N,2^d] | None) of one level - wisp_hashgrid_cells (diagnostic)."""
    coords = _need(coords, torch.float32, "coords")
    n, dim = coords.shape
    cell = torch.empty(n, dim, dtype=torch.int32, device=coords.device)
    frac = torch.empty(n, dim, dtype=torch.float32, device=coords.device)
    corners = torch.empty(n, 1 << dim, dtype=torch.int32, device=coords.device) if with_corners else None
    _check(lib.wisp_hashgrid_cells(_p(coords), n, dim, int(resolution), int(codebook_bitwidth), _p(cell), _p(frac),
                                   _p(corners) if corners is not None else None, _stream()), "wisp_hashgrid_cells")
    return cell, frac, corners


def _bwd_workspace(device, nbytes):
    """Scratch of the binned backward, cached per (device, stream): two backward calls on different streams (the trainer's
    side stream next to a second model) must not share records."""
    key = (device, _stream().value)
    buf = _bwd_ws.get(key)
    if buf is None or buf.numel() < nbytes or buf.numel() > 3 * nbytes + (64 << 20):    # grow, or give an oversized one back
        buf = None
        _bwd_ws[key] = None
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _bwd_ws[key] = buf
    return buf


# ------------------------------------------------------------------------------------------------ scans / packs
def _aligned16(t):
    """the scan kernels read their input with 16-byte loads: a slice that starts mid-allocation is copied."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


def exclusive_scan(counts):
    """int32 counts [n] -> int64 offsets [n+1] (offsets[n] = total)."""
    counts = _aligned16(_need(counts, torch.int32, "counts"))
    n = counts.shape[0]
    offsets = torch.empty(n + 1, dtype=torch.int64, device=counts.device)
    ws = torch.empty(int(lib.wisp_scan_workspace_bytes(n)), dtype=torch.uint8, device=counts.device)
    _check(lib.wisp_exclusive_scan_i32(_p(counts), n, _p(offsets), _p(ws), _stream()), "exclusive_scan")
    return offsets


def inclusive_scan(values):
    """kaolin._C.render.spc.inclusive_sum_cuda (octree_as.py:351): int32 -> int32 inclusive prefix sum."""
    values = _aligned16(_need(values, torch.int32, "values"))
    n = values.shape[0]
    out = torch.empty_like(values)
    ws = torch.empty(int(lib.wisp_scan_workspace_bytes(n)), dtype=torch.uint8, device=values.device)
    _check(lib.wisp_inclusive_scan_i32(_p(values), n, _p(out), _p(ws), _stream()), "inclusive_scan")
    return out


def mark_pack_boundaries(ids):
    """kaolin mark_pack_boundaries / mark_first_hit (octree_as.py:300, :228): bool [n]."""
    ids = _need(ids, None, "ids")
    n = ids.shape[0]
    out = torch.empty(n, dtype=torch.bool, device=ids.device)
    if ids.dtype == torch.int64:
        _check(lib.wisp_mark_pack_boundaries_i64(_p(ids), n, _p(out), _stream()), "mark_pack_boundaries")
    elif ids.dtype == torch.int32:
        _check(lib.wisp_mark_pack_boundaries_i32(_p(ids), n, _p(out), _stream()), "mark_pack_boundaries")
    else:
        raise TypeError(f"mark_pack_boundaries expects int32/int64 ids, got {ids.dtype}")
    return out


def pack_starts(boundary):
    """boundary bool [S] -> int64 [P] start index of every pack (== boundary.nonzero()[:,0])."""
    boundary = _need(boundary, torch.bool, "boundary")
    n = boundary.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=boundary.device)
    ntiles = (n + 2047) // 2048
    counts = torch.empty(ntiles, dtype=torch.int32, device=boundary.device)
    _check(lib.wisp_boundary_tile_counts(_p(boundary), n, _p(counts), _stream()), "boundary_tile_counts")
    offsets = exclusive_scan(counts)
    num_packs = int(offsets[-1].item())
    starts = torch.empty(num_packs, dtype=torch.int64, device=boundary.device)
    if num_packs == 0:
        return starts
    _check(lib.wisp_boundary_pack_starts(_p(boundary), n, _p(offsets), _p(starts), _stream()), "boundary_pack_starts")
    return starts


# ------------------------------------------------------------------------------------------------ SPC
def spc_query(octree, exsum, coords, level, with_parents=False):
    """kaolin.ops.spc.unbatched_query (octree_as.py:162): int64 [Q] or [Q, level+1]."""
    coords = _need(coords, torch.float32, "coords").reshape(-1, 3)
    octree = _need(octree, torch.uint8, "octree")
    exsum = _need(exsum, torch.int32, "exsum")
    n = coords.shape[0]
    shape = (n, level + 1) if with_parents else (n,)
    pidx = torch.empty(shape, dtype=torch.int64, device=coords.device)
    _check(lib.wisp_spc_query(_p(octree), _p(exsum), _p(coords), n, level, int(with_parents), _p(pidx), _stream()),
           "spc_query")
    return pidx


def spc_query_chain(octree, exsum, points, coords, level, first_level, hint=None, hint_group=1):
    """Columns first_level .. level of spc_query(with_parents=True): int64 [Q, level - first_level + 1].  hint (optional): int32
    [ceil(Q / hint_group)] cells of first_level known for groups of hint_group consecutive coordinates (-1 = none) - the
    nuggets a 'voxel' march sampled; they shorten the walk where they are right and change nothing where they are not."""
    coords = _need(coords, torch.float32, "coords").reshape(-1, 3)
    octree = _need(octree, torch.uint8, "octree")
    exsum = _need(exsum, torch.int32, "exsum")
    n = coords.shape[0]
    if hint is not None:
        hint = _need(hint, torch.int32, "hint").reshape(-1)
        assert hint_group >= 1 and hint.shape[0] * hint_group >= n, "one hint per group of coordinates"
        points = _need(points, torch.int16, "points")
    chain = torch.empty(n, level - first_level + 1, dtype=torch.int64, device=coords.device)
    _check(lib.wisp_spc_query_chain(_p(octree), _p(exsum), _p(points if hint is not None else None), _p(coords), n, level,
                                    first_level, _p(hint), int(hint_group), _p(chain), _stream()), "spc_query_chain")
    return chain


def spc_bitfield(level_points, level):
    """Occupancy bits of `level`, row-major cell order (uint32 words viewed as int32 storage)."""
    level_points = _need(level_points, torch.int16, "level_points")
    words = max((8 ** level + 31) // 32, 1)
    bits = torch.empty(words, dtype=torch.int32, device=level_points.device)
    _check(lib.wisp_spc_build_bitfield(_p(level_points), level_points.shape[0], level, _p(bits), _stream()),
           "spc_build_bitfield")
    return bits


SPC_DEVICE_BUILD_MAX_LEVEL = 9          # leaf mask = 8^level bytes (128 MiB at level 9)


def spc_build(level, points=None, leaf_mask=None):
    """Point hierarchy of a level-`level` octree from quantised cells (i16 [n,3], any order, duplicates allowed) or from a
    dense Morton-order occupancy of the finest level (u8 / bool [8^level]) -> (octree u8, points i16 [P,3], pyramid i32
    [2, level+2] on the host, exsum i32 [len(octree)+1]) - what unbatched_points_to_octree + octree_to_spc return, built
    without a sort (csrc/spc.hip: 'SPC build on the device').  Returns None when no cell is occupied."""
    assert 1 <= level <= SPC_DEVICE_BUILD_MAX_LEVEL
    n_leaf = 8 ** level
    n_dense = (8 ** (level + 1) - 1) // 7
    if leaf_mask is not None:
        leaf_mask = _need(leaf_mask, None, "leaf_mask")
        assert leaf_mask.numel() == n_leaf and leaf_mask.dtype in (torch.uint8, torch.bool)
        dev = leaf_mask.device
        dense = torch.empty(n_dense, dtype=torch.uint8, device=dev)
        dense[n_dense - n_leaf:].copy_(leaf_mask.reshape(-1).view(torch.uint8))
    else:
        points = _need(points, torch.int16, "points")
        dev = points.device
        dense = torch.zeros(n_dense, dtype=torch.uint8, device=dev)
        _check(lib.wisp_spc_mask_from_points(_p(points), points.shape[0], level, c_vp(dense.data_ptr() + n_dense - n_leaf),
                                             _stream()), "spc_mask_from_points")
    _check(lib.wisp_spc_dense_bytes(_p(dense), level, _stream()), "spc_dense_bytes")
    starts = pack_starts(dense.view(torch.bool))         # dense position of every point, hierarchy order (one read-back)
    P = starts.shape[0]
    if P == 0:
        return None
    bounds = torch.tensor([(8 ** l - 1) // 7 for l in range(level + 2)], dtype=torch.int64, device=dev)
    first = torch.searchsorted(starts, bounds).cpu()     # first point of every level (second read-back: the pyramid is host data)
    pyramid = torch.zeros(2, level + 2, dtype=torch.int32)
    pyramid[1, :level + 1] = first[:level + 1].int()
    pyramid[1, level + 1] = P
    pyramid[0, :level + 1] = (first[1:] - first[:-1]).int()
    if int(pyramid[0, level]) == 0:
        return None
    num_nodes = int(first[level])
    octree = dense[starts[:num_nodes]]
    pts = torch.empty(P, 3, dtype=torch.int16, device=dev)
    _check(lib.wisp_spc_points_from_index(_p(starts), P, level, _p(pts), _stream()), "spc_points_from_index")
    exsum = torch.zeros(num_nodes + 1, dtype=torch.int32, device=dev)
    exsum[1:] = torch.cumsum(_popc_table(dev)[octree.long()], 0)
    return octree, pts, pyramid, exsum


_POPC = {}


def _popc_table(dev):
    t = _POPC.get(dev)
    if t is None:
        t = _POPC[dev] = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32, device=dev)
    return t


RAYTRACE_CACHE_CAP = 64                  # nuggets per ray parked by the count phase (12 B each)
RAYTRACE_CACHE_LIMIT = 2 << 30           # bytes; beyond that the emit phase walks the octree again


def spc_raytrace_begin(octree, points, exsum, origins, dirs, level):
    """First half of spc_raytrace: per-ray nugget counts (their first RAYTRACE_CACHE_CAP nuggets parked) and the offsets.
    Needs no field parameters and no host read-back - the total travels to pinned memory on a side stream - so a trainer can
    issue it for the NEXT batch early (the 'voxel' / 'uniform' marches' equivalent of raymarch_ray_count)."""
    origins = _need(origins, torch.float32, "origins").reshape(-1, 3)
    dirs = _need(dirs, torch.float32, "dirs").reshape(-1, 3)
    octree = _need(octree, torch.uint8, "octree")
    points = _need(points, torch.int16, "points")
    exsum = _need(exsum, torch.int32, "exsum")
    R = origins.shape[0]
    dev = origins.device
    counts = torch.empty(R, dtype=torch.int32, device=dev)
    # the count phase parks the first RAYTRACE_CACHE_CAP nuggets of every ray, so that emit is a copy, not a second walk
    cap = RAYTRACE_CACHE_CAP if R * RAYTRACE_CACHE_CAP * 12 <= RAYTRACE_CACHE_LIMIT else 0
    cache = torch.empty(R * cap * 3, dtype=torch.float32, device=dev) if cap else None
    _check(lib.wisp_spc_raytrace_count(_p(octree), _p(points), _p(exsum), _p(origins), _p(dirs), R, level, _p(counts),
                                       _p(cache), cap, _stream()), "spc_raytrace_count")
    st = dict(octree=octree, points=points, exsum=exsum, origins=origins, dirs=dirs, level=level, cache=cache, cap=cap,
              offsets=exclusive_scan(counts))
    if dev.type == "cuda":
        _read_total_async(st)
    return st


def spc_raytrace_finish(st, with_exit=False):
    """Second half: size read-back (as the reference's kaolin op does), allocation and emit."""
    origins, offsets = st["origins"], st["offsets"]
    R, dev = origins.shape[0], origins.device
    M = _total(st)
    ridx = torch.empty(M, dtype=torch.int32, device=dev)
    pidx = torch.empty(M, dtype=torch.int32, device=dev)
    depth = torch.empty(M, 2 if with_exit else 1, dtype=torch.float32, device=dev)
    if M:
        _check(lib.wisp_spc_raytrace_emit(_p(st["octree"]), _p(st["points"]), _p(st["exsum"]), _p(origins), _p(st["dirs"]), R,
                                          st["level"], _p(offsets), int(with_exit), _p(st["cache"]), st["cap"], _p(ridx),
                                          _p(pidx), _p(depth), _stream()), "spc_raytrace_emit")
    return ridx, pidx, depth, offsets


def spc_raytrace(octree, points, exsum, origins, dirs, level, with_exit=False):
    """kaolin.render.spc.unbatched_raytrace (octree_as.py:183-185).
    Returns (ridx i32 [M], pidx i32 [M], depth f32 [M,1|2], ray_offsets i64 [R+1])."""
    return spc_raytrace_finish(spc_raytrace_begin(octree, points, exsum, origins, dirs, level), with_exit)


def spc_trilinear_coeffs(coords, voxel_points, level):
    """kaolin coords_to_trilinear_coeffs (codebook_grid.py:164): coords [V,S,3], voxel_points i16 [V,3] -> [V,S,8]."""
    coords = _need(coords, torch.float32, "coords")
    voxel_points = _need(voxel_points, torch.int16, "points").reshape(-1, 3)
    V, S = coords.shape[0], coords.shape[1]
    out = torch.empty(V, S, 8, dtype=torch.float32, device=coords.device)
    _check(lib.wisp_spc_trilinear_coeffs(_p(coords), _p(voxel_points), V, S, level, _p(out), _stream()), "spc_trilinear_coeffs")
    return out


def _pidx_arg(pidx):
    if pidx.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"pidx must be int32 or int64, got {pidx.dtype}")
    return _need(pidx, None, "pidx"), int(pidx.dtype == torch.int64)


def spc_trilinear_forward(coords, pidx, points, trinkets, feats, level, half_round=False):
    """kaolin unbatched_interpolate_trilinear (octree_grid.py:147-149): coords [V,S,3] -> f32 [V,S,C]."""
    coords = _need(coords, torch.float32, "coords")
    pidx, is64 = _pidx_arg(pidx)
    points = _need(points, torch.int16, "points")
    trinkets = _need(trinkets, torch.int32, "trinkets")
    feats = _need(feats, None, "feats")
    V, S, C = coords.shape[0], coords.shape[1], feats.shape[1]
    out = torch.empty(V, S, C, dtype=torch.float32, device=coords.device)
    _check(lib.wisp_spc_trilinear_fwd(_p(coords), _p(pidx), is64, _p(points), _p(trinkets), _p(feats), _DTYPE_CODE[feats.dtype],
                                      V, S, C, level, int(half_round), _p(out), _stream()), "spc_trilinear_fwd")
    return out


_spc_ws = _Scratch(zeroed=True)
_ZEROED_SCRATCH.append(_spc_ws)


def _spc_bwd_workspace(device, total_rows, channels, dict_elems=0):
    """Scratch of the order-free trilinear / codebook backward (wisp_spc_bwd_workspace_bytes): zero when handed out for the first
    time and left zero by every call, so one buffer per (device, stream) serves every grid; it only ever grows (_Scratch: a
    buffer a captured graph points into is never freed or replaced under it; a failed call drops the cache)."""
    need = int(lib.wisp_spc_bwd_workspace_bytes(total_rows, channels, dict_elems))
    if need < 0:
        raise RuntimeError("wisp_spc_bwd_workspace_bytes: bad sizes")
    return _spc_ws.get(device, need)


def _host_i64(values):
    arr = np.ascontiguousarray(np.asarray(values, dtype=np.int64).reshape(-1))
    return arr, arr.ctypes.data_as(c_vp)


def spc_trilinear_backward(coords, pidx, points, trinkets, grad_out, feats_shape, level, out=None):
    """-> gradient of the [rows, C] feature tensor (f32), bitwise reproducible.  out: optional f32 tensor it is ADDED to."""
    coords = _need(coords, torch.float32, "coords")
    pidx, is64 = _pidx_arg(pidx)
    grad_out = _need(grad_out, torch.float32, "grad_out")
    V, S, C = coords.shape[0], coords.shape[1], feats_shape[1]
    grad = torch.zeros(tuple(feats_shape), dtype=torch.float32, device=coords.device) if out is None else _need(out, torch.float32, "out")
    assert tuple(grad.shape) == tuple(feats_shape)
    ws = _spc_bwd_workspace(coords.device, feats_shape[0], C)
    _check(lib.wisp_spc_trilinear_bwd(_p(coords), _p(pidx), is64, _p(_need(points, torch.int16, "points")),
                                      _p(_need(trinkets, torch.int32, "trinkets")), _p(grad_out), V, S, C, level, feats_shape[0],
                                      _p(grad), _p(ws), ws.numel(), _stream()), "spc_trilinear_bwd")
    return grad


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr, ctypes.cast(arr, c_vp)


def spc_trilinear_multi_forward(coords, chain, points, trinkets, feats_list, levels, half_round, sum_lods):
    """All LODs of an OctreeGrid in one launch: coords [N,3], chain i64 [N,L] -> f32 [N, L*C] ('cat') or [N, C] ('sum')."""
    coords = _need(coords, torch.float32, "coords")
    assert chain.is_cuda and chain.dtype == torch.int64 and chain.stride(-1) == 1
    points = _need(points, torch.int16, "points")
    trinkets = _need(trinkets, torch.int32, "trinkets")
    feats_list = [_need(f, None, "feats") for f in feats_list]
    N, L, C = coords.shape[0], len(feats_list), feats_list[0].shape[1]
    assert chain.shape[0] == N and chain.shape[1] >= L and all(f.shape[1] == C and f.dtype == feats_list[0].dtype for f in feats_list)
    out = torch.empty(N, C if sum_lods else L * C, dtype=torch.float32, device=coords.device)
    farr, fptr = _ptr_array(feats_list)
    larr, lptr = _host_i32(levels)
    _check(lib.wisp_spc_trilinear_multi_fwd(_p(coords), _p(chain), chain.stride(0), _p(points), _p(trinkets), fptr,
                                            _DTYPE_CODE[feats_list[0].dtype], N, L, lptr, C, int(half_round), int(sum_lods),
                                            _p(out), _stream()), "spc_trilinear_multi_fwd")
    return out


def spc_trilinear_multi_backward(coords, chain, points, trinkets, grad_out, feats_shapes, levels, sum_lods, out=None):
    """out: optional list of f32 tensors of feats_shapes the corner gradients are ADDED to (e.g. the parameters' .grad).
    One scatter launch for all levels; bitwise reproducible (64-bit fixed-point corner sums)."""
    coords = _need(coords, torch.float32, "coords")
    grad_out = _need(grad_out, torch.float32, "grad_out")
    N, L, C = coords.shape[0], len(feats_shapes), feats_shapes[0][1]
    if out is None:
        grads = [torch.zeros(tuple(sh), dtype=torch.float32, device=coords.device) for sh in feats_shapes]
    else:
        grads = [_need(g, torch.float32, "out") for g in out]
        assert len(grads) == L and all(tuple(g.shape) == tuple(sh) for g, sh in zip(grads, feats_shapes))
    garr, gptr = _ptr_array(grads)
    larr, lptr = _host_i32(levels)
    rarr, rptr = _host_i64([sh[0] for sh in feats_shapes])
    ws = _spc_bwd_workspace(coords.device, int(rarr.sum()), C)
    _check(lib.wisp_spc_trilinear_multi_bwd(_p(coords), _p(chain), chain.stride(0), _p(_need(points, torch.int16, "points")),
                                            _p(_need(trinkets, torch.int32, "trinkets")), _p(grad_out), N, L, lptr, rptr, C,
                                            int(sum_lods), gptr, _p(ws), ws.numel(), _stream()), "spc_trilinear_multi_bwd")
    return grads


def triplane_forward(coords, planes, sum_lods):
    """TriplanarGrid lookup of all levels: coords [N,3], planes = [x0, y0, z0, x1, ...] each [1 or -, fdim, R, R] -> f32
    [N, L*3*fdim] ('cat') or [N, 3*fdim] ('sum')."""
    coords = _need(coords, torch.float32, "coords")
    planes = [_need(p, torch.float32, "plane") for p in planes]
    L = len(planes) // 3
    fdim, sizes = planes[0].shape[-3], [int(planes[3 * l].shape[-1]) for l in range(L)]
    assert len(planes) == 3 * L and all(p.shape[-3] == fdim and p.shape[-1] == p.shape[-2] for p in planes)
    N = coords.shape[0]
    out = torch.empty(N, (1 if sum_lods else L) * 3 * fdim, dtype=torch.float32, device=coords.device)
    parr, pptr = _ptr_array(planes)
    sarr, sptr = _host_i32(sizes)
    _check(lib.wisp_triplane_fwd(_p(coords), N, pptr, sptr, L, fdim, int(sum_lods), _p(out), _stream()), "triplane_fwd")
    return out


def triplane_backward(coords, grad_out, plane_shapes, sum_lods):
    coords = _need(coords, torch.float32, "coords")
    grad_out = _need(grad_out, torch.float32, "grad_out")
    L = len(plane_shapes) // 3
    fdim, sizes = plane_shapes[0][-3], [int(plane_shapes[3 * l][-1]) for l in range(L)]
    grads = [torch.zeros(tuple(sh), dtype=torch.float32, device=coords.device) for sh in plane_shapes]
    garr, gptr = _ptr_array(grads)
    sarr, sptr = _host_i32(sizes)
    _check(lib.wisp_triplane_bwd(_p(coords), coords.shape[0], _p(grad_out), sptr, L, fdim, int(sum_lods), gptr, _stream()),
           "triplane_bwd")
    return grads


CODEBOOK_DECODE_ROWS = os.environ.get("WISP_CODEBOOK_DECODE_ROWS", "1") != "0"


def codebook_trilinear_forward(coords, pidx, points, trinkets, logits, dictionary, level, training):
    """Fused VQAD lookup + trilinear blend: coords [V,S,3] -> f32 [V,S,F]."""
    coords = _need(coords, torch.float32, "coords")
    pidx, is64 = _pidx_arg(pidx)
    logits = _need(logits, torch.float32, "logits")
    dictionary = _need(dictionary, torch.float32, "dictionary")
    V, S = coords.shape[0], coords.shape[1]
    K, F = dictionary.shape
    if CODEBOOK_DECODE_ROWS and V * S >= 4 * logits.shape[0]:
        # decode every logits row once, then the plain trilinear blend (bit-identical; see wisp_codebook_decode_rows)
        return spc_trilinear_forward(coords, pidx, points, trinkets, codebook_decode_rows(logits, dictionary, training), level)
    out = torch.empty(V, S, F, dtype=torch.float32, device=coords.device)
    _check(lib.wisp_codebook_trilinear_fwd(_p(coords), _p(pidx), is64, _p(_need(points, torch.int16, "points")),
                                           _p(_need(trinkets, torch.int32, "trinkets")), _p(logits), _p(dictionary), V, S, K, F,
                                           level, int(training), _p(out), _stream()), "codebook_trilinear_fwd")
    return out


def codebook_decode_rows(logits, dictionary, training):
    """The dictionary vector every logits row selects (straight-through softmax one-hot in training, argmax in eval):
    logits [rows, K], dictionary [K, F] -> f32 [rows, F]; the plain trilinear blend over it equals the fused lookup bit for bit."""
    logits = _need(logits, torch.float32, "logits")
    dictionary = _need(dictionary, torch.float32, "dictionary")
    K, F = dictionary.shape
    decoded = torch.empty(logits.shape[0], F, dtype=torch.float32, device=logits.device)
    _check(lib.wisp_codebook_decode_rows(_p(logits), _p(dictionary), logits.shape[0], K, F, int(training), _p(decoded),
                                         _stream()), "codebook_decode_rows")
    return decoded


def codebook_trilinear_backward(coords, pidx, points, trinkets, logits, dictionary, grad_out, level, out=None):
    """-> (grad logits, grad dictionary), bitwise reproducible.  out: optional pair of f32 tensors the result is ADDED to -
    e.g. the parameters' .grad."""
    coords = _need(coords, torch.float32, "coords")
    pidx, is64 = _pidx_arg(pidx)
    logits = _need(logits, torch.float32, "logits")
    dictionary = _need(dictionary, torch.float32, "dictionary")
    grad_out = _need(grad_out, torch.float32, "grad_out")
    V, S = coords.shape[0], coords.shape[1]
    K, F = dictionary.shape
    if out is None:
        g_logits = torch.zeros_like(logits)
        g_dict = torch.zeros_like(dictionary)
    else:
        g_logits, g_dict = _need(out[0], torch.float32, "out logits"), _need(out[1], torch.float32, "out dictionary")
        assert g_logits.shape == logits.shape and g_dict.shape == dictionary.shape
    ws = _spc_bwd_workspace(coords.device, logits.shape[0], F, K * F)
    _check(lib.wisp_codebook_trilinear_bwd(_p(coords), _p(pidx), is64, _p(_need(points, torch.int16, "points")),
                                           _p(_need(trinkets, torch.int32, "trinkets")), _p(logits), _p(dictionary), _p(grad_out),
                                           V, S, K, F, level, logits.shape[0], _p(g_logits), _p(g_dict), _p(ws), ws.numel(),
                                           _stream()), "codebook_trilinear_bwd")
    return g_logits, g_dict


def codebook_trilinear_multi_backward(coords, chain, points, trinkets, logits_list, dictionaries, grad_out, levels, sum_lods, out=None):
    """All levels of a CodebookOctreeGrid: coords [N,3], chain i64 [N, >= L], grad_out f32 [N, F] ('sum') or [N, L*F] ->
    ([grad logits per level], [grad dictionary per level]); out: optional pair of such lists the result is ADDED to."""
    coords = _need(coords, torch.float32, "coords")
    grad_out = _need(grad_out, torch.float32, "grad_out")
    logits_list = [_need(t, torch.float32, "logits") for t in logits_list]
    dictionaries = [_need(t, torch.float32, "dictionary") for t in dictionaries]
    N, L = coords.shape[0], len(logits_list)
    K, F = dictionaries[0].shape
    assert len(dictionaries) == L and all(tuple(d.shape) == (K, F) for d in dictionaries) and all(t.shape[1] == K for t in logits_list)
    assert chain.is_cuda and chain.dtype == torch.int64 and chain.stride(-1) == 1 and chain.shape[0] == N and chain.shape[1] >= L
    if out is None:
        g_logits = [torch.zeros_like(t) for t in logits_list]
        g_dicts = [torch.zeros_like(t) for t in dictionaries]
    else:
        g_logits = [_need(t, torch.float32, "out logits") for t in out[0]]
        g_dicts = [_need(t, torch.float32, "out dictionary") for t in out[1]]
        assert all(a.shape == b.shape for a, b in zip(g_logits, logits_list)) and all(a.shape == b.shape for a, b in zip(g_dicts, dictionaries))
    larr, lptr = _host_i32(levels)
    rarr, rptr = _host_i64([t.shape[0] for t in logits_list])
    arrs = [_ptr_array(x) for x in (logits_list, dictionaries, g_logits, g_dicts)]
    ws = _spc_bwd_workspace(coords.device, int(rarr.sum()), F, L * K * F)
    _check(lib.wisp_codebook_trilinear_multi_bwd(_p(coords), _p(chain), chain.stride(0), _p(_need(points, torch.int16, "points")),
                                                 _p(_need(trinkets, torch.int32, "trinkets")), arrs[0][1], arrs[1][1], _p(grad_out),
                                                 N, L, lptr, rptr, K, F, int(sum_lods), arrs[2][1], arrs[3][1], _p(ws), ws.numel(),
                                                 _stream()), "codebook_trilinear_multi_bwd")
    return g_logits, g_dicts


# ------------------------------------------------------------------------------------------------ raymarch
def _alloc_samples(S, dev):
    return (torch.empty(S, dtype=torch.int64, device=dev), torch.empty(S, 3, dtype=torch.float32, device=dev),
            torch.empty(S, 1, dtype=torch.float32, device=dev), torch.empty(S, 1, dtype=torch.float32, device=dev),
            torch.empty(S, dtype=torch.bool, device=dev))


RAYMARCH_COARSE_MAX_LEVEL = 5


def raymarch_coarse_level(near, far, num_samples, level):
    """Occupancy level whose cell is about as long as one 64-candidate chunk of a ray (what the count kernel's LDS pre-test
    wants), or None when no coarser level than `level` qualifies."""
    chunk = 64.0 * abs(float(far) - float(near)) / max(int(num_samples), 1)
    if not (chunk > 0.0):
        return None
    lc = min(int(np.floor(np.log2(2.0 / chunk))), RAYMARCH_COARSE_MAX_LEVEL, level - 1)
    return lc if lc >= 1 else None


def raymarch_ray_count(occ_bits, octree, exsum, origins, dirs, near, far, num_samples, level, jitter=None, seed=0,
                       coarse_bits=None, coarse_level=0, read_total_async=True):
    """First half of OctreeAS._raymarch_ray (octree_as.py:247-309): occupancy test of every candidate + per-ray offsets.
    Needs no field parameters and no host read-back, so a trainer can issue it for the NEXT batch early.  Returns the state
    raymarch_ray_finish() expands into packed samples.  coarse_bits / coarse_level: optional bitfield of a coarser level
    of the same octree (see raymarch_coarse_level); it prunes work, never results."""
    origins = _need(origins, torch.float32, "origins").reshape(-1, 3)
    dirs = _need(dirs, torch.float32, "dirs").reshape(-1, 3)
    R, dev = origins.shape[0], origins.device
    if jitter is not None:
        jitter = _need(jitter, torch.float32, "jitter")
        assert jitter.numel() == R * num_samples, "jitter must be [num_rays, num_samples]"
    near32 = float(np.float32(near))
    range32 = float(np.float32(float(far) - float(near)))      # depth *= (dist_max - dist_min), python double -> f32
    words = (num_samples + 31) // 32
    hitmask = torch.empty(R, words, dtype=torch.int32, device=dev)
    counts = torch.empty(R, dtype=torch.int32, device=dev)
    _check(lib.wisp_raymarch_ray_count(_p(occ_bits), _p(octree), _p(exsum), _p(origins), _p(dirs), R, near32, range32,
                                       num_samples, level, _p(jitter), seed, _p(coarse_bits), int(coarse_level), _p(hitmask),
                                       _p(counts), _stream()),
           "raymarch_ray_count")
    offsets = exclusive_scan(counts)
    st = dict(origins=origins, dirs=dirs, near32=near32, range32=range32, num_samples=num_samples, jitter=jitter, seed=seed,
              hitmask=hitmask, offsets=offsets)
    if read_total_async:                 # (a caller that finishes at once reads the total with a plain .item(): see raymarch_ray)
        _read_total_async(st)
    return st


# The sample count has to reach the host before the packed outputs can be allocated.  offsets[-1].item() on the compute
# stream would drain EVERYTHING queued there (with the trainer's one-batch look-ahead: the whole previous step) and leave
# the GPU idle while the host wakes up and launches again (measured: 43 us per step).  Instead the total is copied to
# pinned memory on a side stream that only waits for the scan; raymarch_ray_finish() waits for that copy alone.
def _parse_cpulist(text):
    """'64-127,192-255' (sysfs cpulist) -> set of CPU numbers"""
    cpus = set()
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_host_near_device(index=None):
    """Pin this process's host threads to the CPUs of the GPU's NUMA node (one process per GPU: call it once, before the first
    launch).  An MI355X node has two CPU sockets; a process the scheduler parks on the far one pays the socket hop on every doorbell,
    event and read-back - the GPU-paced fused step does not notice, a host-bound loop does: the unchanged reference trainer's step
    (bench.py's dropin_regime) ran 8 % faster on the mean of ten alternating runs (profiles/r05_numa_binding_dropin.txt).  -> dict(numa_node, cpus) or None when there is nothing to bind to (no sysfs topology, one node, affinity already narrower).
    WISP_NUMA_BIND=0 switches it off."""
    if os.environ.get("WISP_NUMA_BIND", "1") == "0" or not torch.cuda.is_available() or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        p = torch.cuda.get_device_properties(torch.cuda.current_device() if index is None else index)
        base = "/sys/bus/pci/devices/%04x:%02x:%02x.0/" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open(base + "numa_node") as f:
            node = int(f.read().strip())
        with open(base + "local_cpulist") as f:
            text = f.read().strip()
        allowed = os.sched_getaffinity(0)
        cpus = _parse_cpulist(text) & allowed
        if node < 0 or not cpus or cpus == allowed:
            return None
        os.sched_setaffinity(0, cpus)
        return {"numa_node": node, "cpus": len(cpus)}
    except (OSError, ValueError, AttributeError):
        return None


_reader_pool = {}                       # device index -> idle HostReader handles (wisp_host_reader_*: csrc/misc.hip)


class _PendingRead:
    """One issued read-back: owns a HostReader handle until its value is taken - or until the march state holding it is dropped
    (a prune, new rays, an exception between count and finish), in which case the finaliser waits for the 8-byte copy and hands
    the reader back to its device's pool instead of leaking a pinned word and two events."""
    __slots__ = ("dev", "reader")

    def __init__(self, dev, reader):
        self.dev, self.reader = dev, reader

    def take(self):
        reader, self.reader = self.reader, None
        if reader is None:
            raise RuntimeError("read-back already taken")
        value = ctypes.c_int64(0)
        try:
            _check(lib.wisp_host_reader_wait(reader, ctypes.byref(value)), "host_reader_wait")
        finally:
            _reader_pool.setdefault(self.dev, []).append(reader)      # (the handle stays valid after a failed wait)
        return int(value.value)

    def __del__(self):
        reader, self.reader = self.reader, None
        if reader is None:
            return
        try:
            value = ctypes.c_int64(0)
            lib.wisp_host_reader_wait(reader, ctypes.byref(value))       # the copy still targets the reader's pinned word
            _reader_pool.setdefault(self.dev, []).append(reader)
        except Exception:                                                # interpreter shutdown: the library may be gone
            pass


def _read_total_async(st):
    """Issue the read-back of offsets[-1] (the packed sample count): two C calls - event on the compute stream, side stream waits,
    8-byte copy to pinned memory, event.  (Five torch calls until round 5: 25 us of host time per step.)"""
    offsets = st["offsets"]
    cur = torch.cuda.current_device()
    dev = offsets.device.index if offsets.device.index is not None else cur
    pool = _reader_pool.setdefault(dev, [])
    if pool:
        reader = pool.pop()
    elif dev == cur:
        reader = lib.wisp_host_reader_create()
    else:                               # a reader binds the side stream of the device current at creation: make that the tensor's
        with torch.cuda.device(dev):
            reader = lib.wisp_host_reader_create()
    if not reader:
        raise RuntimeError(f"wisp_host_reader_create failed: {last_error()}")
    pending = _PendingRead(dev, reader)                                  # from here on the handle cannot leak
    _check(lib.wisp_host_reader_issue(reader, c_vp(offsets.data_ptr() + 8 * (offsets.numel() - 1)), _stream()), "host_reader_issue")
    st["total_reader"] = pending


def _total(st):
    pending = st.pop("total_reader", None)
    if pending is None:
        return int(st["offsets"][-1].item())
    return pending.take()


def raymarch_ray_finish(st, with_dirs=False):
    """Second half: read the sample count back (the reference syncs here too: nonzero, octree_as.py:288), allocate and
    emit.  Returns (ridx, samples, depth, deltas, boundary, ray_offsets) - plus the per-sample view directions
    dirs[ridx] ([S,3], what packed_rf_tracer.py:120 gathers) when with_dirs is set."""
    origins, offsets = st["origins"], st["offsets"]
    R, dev = origins.shape[0], origins.device
    S = _total(st)
    ridx, samples, depth, deltas, boundary = _alloc_samples(S, dev)
    sample_dirs = torch.empty(S, 3, dtype=torch.float32, device=dev) if with_dirs else None
    if S:
        _check(lib.wisp_raymarch_ray_emit(_p(origins), _p(st["dirs"]), R, st["near32"], st["range32"], st["num_samples"],
                                          _p(st["jitter"]), st["seed"], _p(st["hitmask"]), _p(offsets), _p(ridx), _p(samples),
                                          _p(depth), _p(deltas), _p(boundary), _p(sample_dirs), _stream()), "raymarch_ray_emit")
    if with_dirs:
        return ridx, samples, depth, deltas, boundary, offsets, sample_dirs
    return ridx, samples, depth, deltas, boundary, offsets


def raymarch_ray(occ_bits, octree, exsum, origins, dirs, near, far, num_samples, level, jitter=None, seed=0,
                 coarse_bits=None, coarse_level=0):
    """OctreeAS._raymarch_ray (octree_as.py:247-309).  Returns (ridx, samples, depth, deltas, boundary, ray_offsets)."""
    # count and finish back to back: nothing is queued between the scan and the read-back, so the side-stream copy + two events of
    # the look-ahead path would only add host time (the drop-in trainer's loop is host bound)
    return raymarch_ray_finish(raymarch_ray_count(occ_bits, octree, exsum, origins, dirs, near, far, num_samples, level,
                                                  jitter, seed, coarse_bits, coarse_level, read_total_async=False))


def raymarch_voxel(origins, dirs, nug_ridx, nug_depth, num_samples, jitter=None, seed=0):
    """OctreeAS._raymarch_voxel after the raytrace (octree_as.py:213-245)."""
    origins = _need(origins, torch.float32, "origins").reshape(-1, 3)
    dirs = _need(dirs, torch.float32, "dirs").reshape(-1, 3)
    nug_ridx = _need(nug_ridx, torch.int32, "nug_ridx")
    nug_depth = _need(nug_depth, torch.float32, "nug_depth")
    M = nug_ridx.shape[0]
    if jitter is not None:
        jitter = _need(jitter, torch.float32, "jitter")
        assert jitter.numel() == M * num_samples, "jitter must be [num_nuggets, num_samples]"
    ridx, samples, depth, deltas, boundary = _alloc_samples(M * num_samples, origins.device)
    if M:
        _check(lib.wisp_raymarch_voxel_emit(_p(origins), _p(dirs), _p(nug_ridx), _p(nug_depth), M, num_samples,
                                            _p(jitter), seed, _p(ridx), _p(samples), _p(depth), _p(deltas),
                                            _p(boundary), _stream()), "raymarch_voxel_emit")
    return ridx, samples, depth, deltas, boundary


def raymarch_uniform(origins, dirs, nug_ridx, nug_depth, ray_offsets, scale):
    """OctreeAS._raymarch_uniform after the raytrace (octree_as.py:336-372) + uniform_sample_cuda."""
    origins = _need(origins, torch.float32, "origins").reshape(-1, 3)
    dirs = _need(dirs, torch.float32, "dirs").reshape(-1, 3)
    nug_ridx = _need(nug_ridx, torch.int32, "nug_ridx")
    nug_depth = _need(nug_depth, torch.float32, "nug_depth")
    M, dev = nug_ridx.shape[0], origins.device
    counts = torch.empty(M, dtype=torch.int32, device=dev)
    if M:
        _check(lib.wisp_raymarch_uniform_count(_p(nug_depth), M, float(scale), _p(counts), _stream()),
               "raymarch_uniform_count")
    offsets = exclusive_scan(counts)
    S = int(offsets[-1].item())                      # blocking read-back as in uniform_sample_cuda.cu:76
    ridx, samples, depth, _, boundary = _alloc_samples(S, dev)
    if S:
        _check(lib.wisp_raymarch_uniform_emit(_p(origins), _p(dirs), _p(nug_ridx), _p(nug_depth), M, float(scale),
                                              _p(offsets), _p(ray_offsets), _p(ridx), _p(samples), _p(depth),
                                              _p(boundary), _stream()), "raymarch_uniform_emit")
    return ridx, samples, depth, boundary, offsets


# ------------------------------------------------------------------------------------------------ packed integration
def packed_sum_reduce(feats, starts):
    feats = _need(feats, torch.float32, "feats")
    S, C = feats.shape
    P = starts.shape[0]
    out = torch.empty(P, C, dtype=torch.float32, device=feats.device)
    _check(lib.wisp_packed_sum_reduce(_p(feats), S, C, _p(starts), P, _p(out), _stream()), "packed_sum_reduce")
    return out


def packed_cumsum(feats, starts, exclusive=False, reverse=False):
    feats = _need(feats, torch.float32, "feats")
    S, C = feats.shape
    out = torch.empty_like(feats)
    _check(lib.wisp_packed_cumsum(_p(feats), S, C, _p(starts), starts.shape[0], int(exclusive), int(reverse), _p(out),
                                  _stream()), "packed_cumsum")
    return out


def composite_fwd(color, density, deltas, depths, ridx, starts, num_rays, bg):
    color = _need(color, torch.float32, "color")
    density = _need(density, torch.float32, "density")
    deltas = _need(deltas, torch.float32, "deltas")
    depths = None if depths is None else _need(depths, torch.float32, "depths")
    S, dev = color.shape[0], color.device
    rgb = torch.empty(num_rays, 3, dtype=torch.float32, device=dev)
    alpha = torch.empty(num_rays, 1, dtype=torch.float32, device=dev)
    depth = None if depths is None else torch.empty(num_rays, 1, dtype=torch.float32, device=dev)
    hit = torch.empty(num_rays, dtype=torch.bool, device=dev)
    weights = torch.empty(S, 1, dtype=torch.float32, device=dev)
    bg_arr, bg_ptr = _host_f32(bg)
    num_packs = num_rays if ridx is None else starts.shape[0]
    if ridx is None:
        assert starts.shape[0] == num_rays + 1, "ray-offset mode needs offsets [num_rays + 1]"
    _check(lib.wisp_composite_fwd(_p(color), _p(density), _p(deltas), _p(depths), _p(ridx), _p(starts), num_packs,
                                  S, num_rays, bg_ptr, _p(rgb), _p(alpha), _p(depth), _p(hit), _p(weights), _stream()),
           "composite_fwd")
    return rgb, alpha, depth, hit, weights


def composite_bwd(grad_rgb, grad_alpha, grad_depth, color, density, deltas, depths, ridx, starts, bg):
    S, dev = color.shape[0], color.device
    grad_rgb = _need(grad_rgb, torch.float32, "grad_rgb")
    grad_alpha = None if grad_alpha is None else _need(grad_alpha, torch.float32, "grad_alpha")
    grad_depth = None if grad_depth is None else _need(grad_depth, torch.float32, "grad_depth")
    g_color = torch.empty(S, 3, dtype=torch.float32, device=dev)
    g_density = torch.empty(S, 1, dtype=torch.float32, device=dev)
    bg_arr, bg_ptr = _host_f32(bg)
    num_packs = starts.shape[0] - 1 if ridx is None else starts.shape[0]
    _check(lib.wisp_composite_bwd(_p(grad_rgb), _p(grad_alpha), _p(grad_depth), _p(color), _p(density), _p(deltas),
                                  _p(depths), _p(ridx), _p(starts), num_packs, S, bg_ptr, _p(g_color),
                                  _p(g_density), _stream()), "composite_bwd")
    return g_color, g_density


def find_depth_bound(query, curr_idxes, nug_depth):
    """wisp._C.render.find_depth_bound_cuda (find_depth_bound.cpp:23-36): int32 [P]."""
    query = _need(query, torch.float32, "query")
    curr_idxes = _need(curr_idxes, torch.int32, "curr_idxes")
    nug_depth = _need(nug_depth, torch.float32, "nug_depth")
    P = query.shape[0]
    out = torch.empty(P, dtype=torch.int32, device=query.device)
    _check(lib.wisp_find_depth_bound(_p(query), _p(curr_idxes), _p(nug_depth), P, nug_depth.shape[0], _p(out), _stream()),
           "find_depth_bound")
    return out


def sphere_trace_step(nug_o, nug_d, nug_depth, nug_pidx, dist_max, thr_close, thr_avg, t, dist, dist_prev, mask, hit,
                      curr_in, curr_out, curr_pidx, x):
    """One fused marching iteration (packed_sdf_tracer.py:118-146); all state tensors are updated in place."""
    P = t.shape[0]
    _check(lib.wisp_sphere_trace_step(P, _p(nug_o), _p(nug_d), _p(nug_depth), _p(nug_pidx), float(np.float32(dist_max)),
                                      float(np.float32(thr_close)), float(np.float32(thr_avg)), _p(t), _p(dist), _p(dist_prev),
                                      _p(mask), _p(hit), _p(curr_in), _p(curr_out), _p(curr_pidx), _p(x), _stream()),
           "sphere_trace_step")


def sdf_trace_step_fused(first, nug_o, nug_d, nug_depth, nug_pidx, dist_max, thr_close, thr_avg, t, dist, dist_prev, mask, hit,
                         curr_in, curr_out, curr_pidx, x, octree, exsum, points, trinkets, feats, levels, half_round, w1, b1, w2, b2,
                         scale, any_active=None):
    """sphere_trace_step + the NeuralSDF / OctreeGrid field query at the new positions, one launch (csrc/spc_interp.hip)."""
    P = nug_o.shape[0]
    n = len(feats)
    for f in feats:
        assert f.is_cuda and f.is_contiguous() and f.dtype == feats[0].dtype and f.shape[1] == feats[0].shape[1]
    fp = (ctypes.c_void_p * n)(*[f.data_ptr() for f in feats])
    lv = (ctypes.c_int32 * n)(*[int(l) for l in levels])
    for a in (w1, b1, w2, b2):
        assert a.is_cuda and a.dtype == torch.float32 and a.is_contiguous()
    _check(lib.wisp_sdf_trace_step_fused(P, int(first), _p(nug_o), _p(nug_d), _p(nug_depth), _p(nug_pidx), float(np.float32(dist_max)),
                                         float(np.float32(thr_close)), float(np.float32(thr_avg)), _p(t), _p(dist), _p(dist_prev),
                                         _p(mask), _p(hit), _p(curr_in), _p(curr_out), _p(curr_pidx), _p(x), _p(octree), _p(exsum),
                                         _p(points), _p(trinkets), fp, _DTYPE_CODE[feats[0].dtype], lv, n, feats[0].shape[1],
                                         int(half_round), _p(w1), _p(b1), _p(w2), _p(b2), w1.shape[0], float(np.float32(scale)),
                                         _p(any_active), _stream()), "sdf_trace_step_fused")


_sdf_scratch = _Scratch(zeroed=False)      # (SDFTrainStep.capture bakes its address into a HIP graph: _Scratch keeps it alive)


def sdf_train_step(coords, gts, octree, exsum, points, trinkets, feats, levels, half_round, w1, b1, w2, b2, grad_feats, grad_w1,
                   grad_b1, grad_w2, grad_b2):
    """Forward + loss + backward of one SDF regression step (wisp_sdf_train_step): coords [n,3], gts [n,1] -> loss f32 [1]
    (= sum((pred - gt)^2) / n); the gradients are ADDED to grad_feats (list, one per level) and grad_w1 / b1 / w2 / b2."""
    coords = _need(coords, torch.float32, "coords")
    gts = _need(gts, torch.float32, "gts").reshape(-1)
    n, L, C, H = coords.shape[0], len(feats), feats[0].shape[1], w1.shape[0]
    dev = coords.device
    assert gts.shape[0] == n and all(f.dtype == torch.float32 and f.is_contiguous() and f.shape[1] == C for f in feats)
    assert all(g.dtype == torch.float32 and g.is_contiguous() and g.shape == f.shape for g, f in zip(grad_feats, feats))
    need = int(lib.wisp_sdf_train_scratch_bytes(n, L, C, H))
    scratch = _sdf_scratch.get(dev, need)
    rarr, rptr = _host_i64([f.shape[0] for f in feats])
    ws = _spc_bwd_workspace(dev, int(rarr.sum()), C)
    farr, fptr = _ptr_array(feats)
    garr, gptr = _ptr_array(grad_feats)
    larr, lptr = _host_i32(levels)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    _check(lib.wisp_sdf_train_step(_p(coords), _p(gts), n, _p(_need(octree, torch.uint8, "octree")), _p(_need(exsum, torch.int32, "exsum")),
                                   _p(_need(points, torch.int16, "points")), _p(_need(trinkets, torch.int32, "trinkets")), fptr, lptr,
                                   rptr, L, C, int(half_round), _p(w1), _p(b1), _p(w2), _p(b2), H, gptr, _p(grad_w1), _p(grad_b1),
                                   _p(grad_w2), _p(grad_b2), _p(loss), _p(scratch), scratch.numel(), _p(ws), ws.numel(), _stream()),
           "sdf_train_step")
    return loss


def small_decoder_forward(x, w1, b1, w2, b2):
    """[n, in] -> [n, 1]: W2 relu(W1 x + b1) + b2 in one launch (csrc/spc_interp.hip)."""
    x = _need(x, torch.float32, "x")
    n, in_dim = x.shape
    out = torch.empty(n, 1, dtype=torch.float32, device=x.device)
    _check(lib.wisp_small_decoder_fwd(_p(x), n, in_dim, w1.shape[0], _p(w1), _p(b1), _p(w2), _p(b2), _p(out), _stream()),
           "small_decoder_fwd")
    return out


def small_decoder_backward(x, w1, b1, w2, b2, grad_out):
    """-> (grad_x [n, in], grad_w1, grad_b1, grad_w2 [hidden], grad_b2 [1])."""
    x = _need(x, torch.float32, "x")
    grad_out = _need(grad_out, torch.float32, "grad_out").reshape(-1)
    n, in_dim = x.shape
    gx = torch.empty_like(x)
    gw1, gb1, gw2, gb2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros_like(b2)
    _check(lib.wisp_small_decoder_bwd(_p(x), n, in_dim, w1.shape[0], _p(w1), _p(b1), _p(w2), _p(b2), _p(grad_out), _p(gx),
                                      _p(gw1), _p(gb1), _p(gw2), _p(gb2), _stream()), "small_decoder_bwd")
    return gx, gw1, gb1, gw2, gb2


_LOSS_KIND = {"huber": 0, "l2": 1, "l1": 2}
_loss_ws = {}


def rgb_loss(rgb, gts, kind):
    """(loss [1], d loss / d rgb) for the mean huber / l2 / l1 error (multiview_trainer.py:140-154)."""
    rgb = _need(rgb, torch.float32, "rgb")
    gts = _need(gts, torch.float32, "gts")
    assert rgb.shape == gts.shape
    grad = torch.empty_like(rgb)
    ws = _loss_ws.get(rgb.device)
    if ws is None:
        ws = _loss_ws[rgb.device] = torch.zeros(257, dtype=torch.float32, device=rgb.device)   # ticket + partials
    loss = torch.empty(1, dtype=torch.float32, device=rgb.device)
    _check(lib.wisp_rgb_loss(_p(rgb), _p(gts), rgb.numel(), _LOSS_KIND[kind], _p(grad), _p(loss), _p(ws), _stream()),
           "rgb_loss")
    return loss, grad


_fused_ws = {}


def composite_loss(color, density, deltas, ray_offsets, num_rays, bg, gts, kind, with_rgb=False):
    """Compositing + rgb loss + compositing backward of a training step in one launch (see wisp_composite_loss):
    -> (loss [1], d loss / d color [S,3], d loss / d density [S,1], rgb [R,3] or None)."""
    color = _need(color, torch.float32, "color")
    density = _need(density, torch.float32, "density")
    deltas = _need(deltas, torch.float32, "deltas")
    gts = _need(gts, torch.float32, "gts")
    ray_offsets = _need(ray_offsets, torch.int64, "ray_offsets")
    S, dev = color.shape[0], color.device
    assert ray_offsets.shape[0] == num_rays + 1 and gts.numel() == num_rays * 3
    g_color = torch.empty(S, 3, dtype=torch.float32, device=dev)
    g_density = torch.empty(S, 1, dtype=torch.float32, device=dev)
    rgb = torch.empty(num_rays, 3, dtype=torch.float32, device=dev) if with_rgb else None
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    key = (dev, _stream().value)                         # per stream: two trainers on one device must not share the partials
    ws = _fused_ws.get(key)
    if ws is None or ws.numel() < num_rays:                                                      # one partial sum of the loss per ray
        ws = _fused_ws[key] = torch.empty(max(8192, 2 * num_rays), dtype=torch.float32, device=dev)
    bg_arr, bg_ptr = _host_f32(bg)
    _check(lib.wisp_composite_loss(_p(color), _p(density), _p(deltas), _p(ray_offsets), num_rays, S, bg_ptr, _p(gts),
                                   _LOSS_KIND[kind], _p(g_color), _p(g_density), _p(rgb), _p(loss), _p(ws), ws.numel(), _stream()),
           "composite_loss")
    return loss, g_color, g_density, rgb


def generate_rays(pixel_x, pixel_y, ortho, x0, y0, width, height, scale_x, scale_y, view_rotation, view_translation):
    """(origins [N,3], dirs [N,3]) for pixel coordinates [N] of one camera (raygen.py:40-119)."""
    pixel_x = _need(pixel_x, torch.float32, "pixel_x").reshape(-1)
    pixel_y = _need(pixel_y, torch.float32, "pixel_y").reshape(-1)
    n, dev = pixel_x.shape[0], pixel_x.device
    assert pixel_y.shape[0] == n
    origins = torch.empty(n, 3, dtype=torch.float32, device=dev)
    dirs = torch.empty(n, 3, dtype=torch.float32, device=dev)
    r_arr, r_ptr = _host_f32(view_rotation)
    t_arr, t_ptr = _host_f32(view_translation)
    assert r_arr.size == 9 and t_arr.size == 3
    f32 = lambda v: float(np.float32(v))
    _check(lib.wisp_generate_rays(_p(pixel_x), _p(pixel_y), n, int(bool(ortho)), f32(x0), f32(y0), f32(width), f32(height),
                                  f32(scale_x), f32(scale_y), r_ptr, t_ptr, _p(origins), _p(dirs), _stream()),
           "generate_rays")
    return origins, dirs


# ------------------------------------------------------------------------------------------------ optimizer
def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0,
               zero_grad=False, bf16_shadow=None):
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    if bf16_shadow is not None:
        assert bf16_shadow.dtype == torch.bfloat16 and bf16_shadow.numel() == param.numel() and bf16_shadow.is_contiguous()
    _check(lib.wisp_adamw_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, beta1, beta2, eps,
                               weight_decay, step, grad_scale, int(zero_grad), _p(bf16_shadow), _stream()), "adamw_step")


def gather_rows(index, tensors, out=None):
    """[t[index] for t in tensors] for up to 4 fp32 tensors with the same number of rows - one launch, one read of `index`
    (the ray-batch sampling of SampleRays, ray_sampler.py:25-35)."""
    index = _need(index, torch.int64, "index").reshape(-1)
    assert 1 <= len(tensors) <= 4
    rows = tensors[0].shape[0]
    srcs, outs = [], []
    for k, t in enumerate(tensors):
        t = _need(t, torch.float32, "tensor")
        assert t.shape[0] == rows and t.is_contiguous()
        srcs.append(t)
        shape = (index.shape[0],) + tuple(t.shape[1:])
        if out is None:
            outs.append(torch.empty(shape, dtype=torch.float32, device=t.device))
        else:                                       # `out`: tensors to gather into (e.g. the static inputs of a captured graph)
            o = out[k]
            assert o.is_cuda and o.dtype == torch.float32 and o.is_contiguous() and tuple(o.shape) == shape
            outs.append(o)
    n = len(srcs)
    widths = [int(t.numel() // max(rows, 1)) for t in srcs]
    src_p = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
    dst_p = (ctypes.c_void_p * n)(*[t.data_ptr() for t in outs])
    w_p = (ctypes.c_int * n)(*widths)
    _check(lib.wisp_gather_rows(_p(index), index.shape[0], rows, n, src_p, w_p, dst_p, _stream()), "gather_rows")
    return outs


ADAMW_MAX_GROUPS = 4            # misc.hip


def adamw_step_groups(param, grad, exp_avg, exp_avg_sq, groups, beta1, beta2, eps, step, grad_scale=1.0, zero_grad=False):
    """One launch for several parameter groups of a flat buffer.  groups: list of (begin, length, lr, weight_decay,
    bf16_shadow or None); a group whose `begin` is a multiple of 4 elements is processed 16 bytes at a time.  More than
    ADAMW_MAX_GROUPS groups go out as several launches."""
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    if len(groups) > ADAMW_MAX_GROUPS:
        for k in range(0, len(groups), ADAMW_MAX_GROUPS):
            adamw_step_groups(param, grad, exp_avg, exp_avg_sq, groups[k:k + ADAMW_MAX_GROUPS], beta1, beta2, eps, step, grad_scale, zero_grad)
        return
    n = len(groups)
    begin = (ctypes.c_int64 * n)(*[int(g[0]) for g in groups])
    length = (ctypes.c_int64 * n)(*[int(g[1]) for g in groups])
    lr = (ctypes.c_float * n)(*[float(g[2]) for g in groups])
    wd = (ctypes.c_float * n)(*[float(g[3]) for g in groups])
    for g in groups:
        assert g[0] + g[1] <= param.numel()
        if g[4] is not None:
            assert g[4].dtype == torch.bfloat16 and g[4].numel() == g[1] and g[4].is_contiguous()
    shadow = (ctypes.c_void_p * n)(*[(g[4].data_ptr() if g[4] is not None else None) for g in groups])
    _check(lib.wisp_adamw_step_groups(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, begin, length, lr, wd, shadow,
                                      beta1, beta2, eps, step, grad_scale, int(zero_grad), _stream()), "adamw_step_groups")


OPTIM_KINDS = {"adamw": 0, "adam": 1, "rmsprop": 2}


def optim_step_groups(kind, param, grad, state1, state2, groups, hyper0, hyper1, eps, step, grad_scale=1.0, zero_grad=False):
    """torch.optim.{AdamW, Adam, RMSprop} over several parameter groups of a flat buffer in one launch (see
    wisp_optim_step_groups).  `groups` as in adamw_step_groups; state1 may be None for RMSprop without momentum."""
    for t in (param, grad, state2) + ((state1,) if state1 is not None else ()):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    n = len(groups)
    begin = (ctypes.c_int64 * n)(*[int(g[0]) for g in groups])
    length = (ctypes.c_int64 * n)(*[int(g[1]) for g in groups])
    lr = (ctypes.c_float * n)(*[float(g[2]) for g in groups])
    wd = (ctypes.c_float * n)(*[float(g[3]) for g in groups])
    for g in groups:
        assert g[0] + g[1] <= param.numel()
        if g[4] is not None:
            assert g[4].dtype == torch.bfloat16 and g[4].numel() == g[1] and g[4].is_contiguous()
    shadow = (ctypes.c_void_p * n)(*[(g[4].data_ptr() if g[4] is not None else None) for g in groups])
    _check(lib.wisp_optim_step_groups(OPTIM_KINDS[kind], _p(param), _p(grad), _p(state1), _p(state2), n, begin, length, lr, wd,
                                      shadow, hyper0, hyper1, eps, step, grad_scale, int(zero_grad), _stream()),
           "optim_step_groups")


# ------------------------------------------------------------------------------------------------ fused NeRF decoder
def _check_decoder_shapes(feats, params, in_dim, hidden, view_freqs):
    if feats.dim() != 2 or feats.shape[1] != in_dim:
        raise ValueError(f"feats must be [S, {in_dim}], got {tuple(feats.shape)}")
    want = int(lib.wisp_nerf_mlp_param_count(in_dim, hidden, view_freqs))
    if params.numel() != want:
        raise ValueError(f"packed decoder parameters: expected {want} floats for in_dim={in_dim}, got {params.numel()}")


def nerf_mlp_forward(feats, dirs, params, in_dim, hidden, view_freqs, compute_bf16, ray_code=None):
    """(rgb [S,3], density [S,1]) = fused density + colour decoders (nerf.py:245-264).
    ray_code = (ridx int64 [S], code from nerf_mlp_dir_code) replaces the per-sample `dirs` (then None)."""
    feats = _need(feats, None, "feats")
    params = _need(params, torch.float32, "params")
    _check_decoder_shapes(feats, params, in_dim, hidden, view_freqs)
    S = feats.shape[0]
    rgb = torch.empty(S, 3, dtype=torch.float32, device=feats.device)
    density = torch.empty(S, 1, dtype=torch.float32, device=feats.device)
    if ray_code is not None:
        if not compute_bf16:
            raise RuntimeError("nerf_mlp_forward: per-ray view codes exist for the bf16-compute kernels only (nerf_mlp_rays_supported)")
        ridx, code = _need(ray_code[0], torch.int64, "ridx"), _need(ray_code[1], torch.bfloat16, "dir_code")
        if ridx.shape[0] != S:
            raise ValueError("ridx must have one entry per sample")
        with _timed("nerf_mlp_fwd", S):
            _check(lib.wisp_nerf_mlp_fwd_rays(_p(feats), _DTYPE_CODE[feats.dtype], _p(code), _p(ridx), S, in_dim, hidden, view_freqs,
                                              _p(params), _p(rgb), _p(density), _stream()), "nerf_mlp_fwd_rays")
        return rgb, density
    dirs = _need(dirs, torch.float32, "dirs")
    with _timed("nerf_mlp_fwd", S):
        _check(lib.wisp_nerf_mlp_fwd(_p(feats), _DTYPE_CODE[feats.dtype], _p(dirs), S, in_dim, hidden, view_freqs, _p(params),
                                     BF16 if compute_bf16 else F32, _p(rgb), _p(density), _stream()), "nerf_mlp_fwd")
    return rgb, density


def nerf_mlp_dir_code(ray_dirs, view_freqs=4):
    """bf16 [R, 32] view code of every ray for nerf_mlp_forward / nerf_mlp_backward(..., ray_code=(ridx, code))."""
    ray_dirs = _need(ray_dirs, torch.float32, "ray_dirs")
    R = ray_dirs.shape[0]
    code = torch.empty(R, 32, dtype=torch.bfloat16, device=ray_dirs.device)
    _check(lib.wisp_nerf_mlp_dir_code(_p(ray_dirs), R, view_freqs, _p(code), _stream()), "nerf_mlp_dir_code")
    return code


def nerf_mlp_rays_supported(feats_dtype, in_dim, hidden, view_freqs, compute_bf16):
    """the per-ray view code path: the bf16-compute hidden-64 kernels, any feature width they take, any I/O type"""
    return (bool(compute_bf16) and 1 <= in_dim <= 32 and hidden == 64 and view_freqs == 4
            and feats_dtype in (torch.float32, torch.float16, torch.bfloat16))


def nerf_mlp_rays_preferred(feats_dtype, in_dim, hidden, view_freqs, compute_bf16):
    """where a trainer should hand the decoder per-ray view codes instead of per-sample directions: wherever the kernels take them.
    On the 32-wide 16-bit rows of the hash-grid field the kernels themselves get faster (forward 70 -> 59 us, backward -6 % at
    2 M samples); on the narrow fp32 rows of the octree / codebook fields they take the same time (A/B on one box, VQAD bench:
    backward 0.311 / 0.312 ms, forward 0.068 / 0.067 ms - the in-kernel encoding is not what those kernels wait for) and the step
    saves the per-sample direction gather: 1.069 -> 1.060 ms.  WISP_MLP_RAYS=wide restricts it to the 32-wide 16-bit rows."""
    ok = nerf_mlp_rays_supported(feats_dtype, in_dim, hidden, view_freqs, compute_bf16)
    if os.environ.get("WISP_MLP_RAYS") == "wide":
        return ok and in_dim == 32 and feats_dtype in (torch.float16, torch.bfloat16)
    return ok


_mlp_workspace = {}


def nerf_mlp_backward(feats, dirs, params, grad_rgb, grad_density, in_dim, hidden, view_freqs, compute_bf16, grad_params=None,
                      ray_code=None):
    """(grad_feats [S,in_dim] in feats.dtype, grad_params fp32 - accumulated into `grad_params` when given)."""
    feats = _need(feats, None, "feats")
    if ray_code is None:
        dirs = _need(dirs, torch.float32, "dirs")
    params = _need(params, torch.float32, "params")
    grad_rgb = _need(grad_rgb, torch.float32, "grad_rgb")
    grad_density = _need(grad_density, torch.float32, "grad_density")
    _check_decoder_shapes(feats, params, in_dim, hidden, view_freqs)
    S, dev = feats.shape[0], feats.device
    grad_feats = torch.empty_like(feats)
    if grad_params is None:
        grad_params = torch.zeros_like(params)
    need = int(lib.wisp_nerf_mlp_bwd_workspace_bytes(S, hidden))
    key = (dev, hidden, _stream().value)          # per stream, like the hash-grid scratch: two backward calls on different streams
    ws = _mlp_workspace.get(key)                  # must not share partial gradient rows
    if ws is None or ws.numel() * 4 < need:
        _mlp_workspace[key] = None
        ws = _mlp_workspace[key] = torch.empty((need + 3) // 4 + 64, dtype=torch.float32, device=dev)
    if ray_code is not None:
        if not compute_bf16:
            raise RuntimeError("nerf_mlp_backward: per-ray view codes exist for the bf16-compute kernels only (nerf_mlp_rays_supported)")
        ridx, code = _need(ray_code[0], torch.int64, "ridx"), _need(ray_code[1], torch.bfloat16, "dir_code")
        if ridx.shape[0] != S:
            raise ValueError("ridx must have one entry per sample")
        with _timed("nerf_mlp_bwd", S):
            _check(lib.wisp_nerf_mlp_bwd_rays(_p(feats), _DTYPE_CODE[feats.dtype], _p(code), _p(ridx), S, in_dim, hidden, view_freqs,
                                              _p(params), _p(grad_rgb), _p(grad_density), _p(grad_feats), _p(grad_params), _p(ws),
                                              ws.numel() * 4, _stream()), "nerf_mlp_bwd_rays")
        return grad_feats, grad_params
    with _timed("nerf_mlp_bwd", S):
        _check(lib.wisp_nerf_mlp_bwd(_p(feats), _DTYPE_CODE[feats.dtype], _p(dirs), S, in_dim, hidden, view_freqs, _p(params),
                                     BF16 if compute_bf16 else F32, _p(grad_rgb), _p(grad_density), _p(grad_feats),
                                     _p(grad_params), _p(ws), ws.numel() * 4, _stream()), "nerf_mlp_bwd")
    return grad_feats, grad_params


# ------------------------------------------------------------------------------------------------ reference-named surface
# The reference's pybind module is `wisp._C` with submodules `ops` / `render` (wisp/csrc/bindings.cpp:21-35); its Python
# callers (wisp/ops/grid.py:92,117; wisp/accelstructs/octree_as.py:353; wisp/ops/geometric.py:22) use exactly the names and
# positional signatures below, so reference code binds to this module unchanged.  Each function is a thin adapter onto
# the C-ABI entry points above.
class _Namespace:
    def __init__(self, name, **fns):
        self.__name__ = name
        self.__dict__.update(fns)


def _resolution_list(resolution):
    # the reference passes a HOST int64 [L,1] tensor and reads it with .item<int>() (hashgrid_interpolate_cuda.cu:365)
    if torch.is_tensor(resolution):
        return [int(r) for r in resolution.reshape(-1).tolist()]
    return [int(r) for r in resolution]


def _ref_hashgrid_interpolate_cuda(coords, codebook, codebook_first_idx, resolution, codebook_bitwidth):
    """hashgrid_interpolate.h:18-23 -> feats [N, L*F] in the dtype of `codebook`."""
    coords = coords.reshape(-1, coords.shape[-1])
    return hashgrid_interpolate(coords, codebook, codebook_first_idx, _resolution_list(resolution), int(codebook_bitwidth))


def _ref_hashgrid_interpolate_backward_cuda(coords, grad_output, codebook, codebook_first_idx, resolution, codebook_bitwidth,
                                            feature_dim, require_grad_coords):
    """hashgrid_interpolate.h:25-33 -> [grad_coords (empty [0] unless requested), grad_codebook in codebook's dtype]."""
    assert int(feature_dim) == codebook.shape[-1]
    coords = coords.reshape(-1, coords.shape[-1])
    grad_output = grad_output.reshape(coords.shape[0], -1)
    grad = hashgrid_interpolate_backward(coords, grad_output, tuple(codebook.shape), codebook_first_idx, _resolution_list(resolution),
                                         int(codebook_bitwidth))
    if require_grad_coords:                     # the reference's arithmetic, as is (hashgrid_interpolate_cuda.cu:163-196)
        grad_coords = hashgrid_grad_coords(coords, grad_output, codebook, codebook_first_idx, _resolution_list(resolution),
                                           int(codebook_bitwidth))
    else:
        grad_coords = torch.empty(0, dtype=torch.float32, device=coords.device)
    return [grad_coords, grad.to(codebook.dtype)]


def _ref_uniform_sample_cuda(scale, ridx, depth, insum):
    """uniform_sample.cpp:28-42 -> [ridx i64 [S], depth_samples f32 [S,1], boundary bool [S]]."""
    ridx = _need(ridx, torch.int32, "ridx")
    depth = _need(depth, torch.float32, "depth")
    insum = _need(insum, torch.int32, "insum")
    V, dev = ridx.shape[0], ridx.device
    S = int(insum[-1].item()) if V else 0                # the reference's blocking cudaMemcpy (uniform_sample_cuda.cu:76)
    new_ridx = torch.empty(S, dtype=torch.int64, device=dev)
    depth_samples = torch.empty(S, 1, dtype=torch.float32, device=dev)
    boundary = torch.empty(S, dtype=torch.bool, device=dev)
    if S:
        _check(lib.wisp_uniform_sample(int(scale), _p(ridx), _p(depth), _p(insum), V, _p(new_ridx), _p(depth_samples),
                                       _p(boundary), _stream()), "uniform_sample")
    return [new_ridx, depth_samples, boundary]


def _ref_find_depth_bound_cuda(query, curr_idxes, depth):
    """find_depth_bound.cpp:23-36: query f32 [P,1], curr_idxes i32 [P], depth f32 [M,2] -> i32 [P]."""
    return find_depth_bound(query.reshape(-1), curr_idxes, depth)


def _ref_grid_interpolate_cuda(coords, feats_in):
    """grid_interpolate.h: coords f32 [N,3] local in [0,1], feats [N,8,F] -> [N,F] in feats' dtype."""
    coords = _need(coords, torch.float32, "coords")
    feats_in = _need(feats_in, None, "feats_in")
    N, F = coords.shape[0], feats_in.shape[-1]
    out = torch.empty(N, F, dtype=feats_in.dtype, device=coords.device)
    _check(lib.wisp_grid_interpolate_fwd(_p(coords), _p(feats_in), _DTYPE_CODE[feats_in.dtype], N, F, _p(out), _stream()),
           "grid_interpolate_fwd")
    return out


def _ref_grid_interpolate_backward_cuda(coords, grad_output, feature_dim):
    """-> grad_feats [N,8,F] in grad_output's dtype."""
    coords = _need(coords, torch.float32, "coords")
    grad_output = _need(grad_output, None, "grad_output")
    N = coords.shape[0]
    grad = torch.empty(N, 8, int(feature_dim), dtype=grad_output.dtype, device=coords.device)
    _check(lib.wisp_grid_interpolate_bwd(_p(coords), _p(grad_output), _DTYPE_CODE[grad_output.dtype], N, int(feature_dim),
                                         _p(grad), _stream()), "grid_interpolate_bwd")
    return grad


def _table_pointers(tables):
    return (ctypes.c_void_p * len(tables))(*[t.data_ptr() for t in tables])


def hashgrid_query(coords, codebooks, resolutions, codebook_bitwidth, probe_bitwidth=0):
    """Eight un-blended corner rows per level (hashgrid_query_cuda.cu:19-66): coords f32 [N,3], one [2^bw, F] table per level
    -> [N, 8, L * P * F] in the tables' dtype, P = 2^probe_bitwidth."""
    coords = _need(coords, torch.float32, "coords")
    tables = [_need(t, codebooks[0].dtype, "codebook") for t in codebooks]
    res = [int(r) for r in resolutions]
    if coords.shape[-1] != 3 or len(tables) != len(res):
        raise ValueError("hashgrid_query: 3-D coordinates and one codebook per resolution")
    N, L, F, P = coords.shape[0], len(res), tables[0].shape[1], 1 << int(probe_bitwidth)
    if any(tuple(t.shape) != (1 << int(codebook_bitwidth), F) for t in tables):
        raise ValueError("hashgrid_query: every codebook is [2^codebook_bitwidth, feature_dim]")
    feats = torch.empty(N, 8, L * P * F, dtype=tables[0].dtype, device=coords.device)
    rarr = (ctypes.c_int32 * L)(*res)
    _check(lib.wisp_hashgrid_query_fwd(_p(coords), N, _table_pointers(tables), _DTYPE_CODE[tables[0].dtype], F, rarr, L,
                                       int(codebook_bitwidth), int(probe_bitwidth), _p(feats), _stream()), "hashgrid_query_fwd")
    return feats


def hashgrid_query_backward(coords, grad_output, resolutions, codebook_rows, codebook_bitwidth, feature_dim, probe_bitwidth=0):
    """-> list of gradient tables [rows_l, feature_dim] in grad_output's dtype (hashgrid_query.cpp:69-97)."""
    coords = _need(coords, torch.float32, "coords")
    grad_output = _need(grad_output, None, "grad_output")
    res = [int(r) for r in resolutions]
    N, L, F = coords.shape[0], len(res), int(feature_dim)
    rows = [int(r) for r in codebook_rows]
    if any(r < (1 << int(codebook_bitwidth)) for r in rows):
        raise ValueError("hashgrid_query_backward: gradient tables need 2^codebook_bitwidth rows")
    if grad_output.numel() != N * 8 * L * (1 << int(probe_bitwidth)) * F:
        raise ValueError("hashgrid_query_backward: grad_output is [N, 8, L * P * F]")
    grads = [torch.zeros(r, F, dtype=grad_output.dtype, device=coords.device) for r in rows]
    rarr = (ctypes.c_int32 * L)(*res)
    _check(lib.wisp_hashgrid_query_bwd(_p(coords), N, _p(grad_output), _DTYPE_CODE[grad_output.dtype], F, rarr, L,
                                       int(codebook_bitwidth), int(probe_bitwidth), _table_pointers(grads), _stream()),
           "hashgrid_query_bwd")
    return grads


def _ref_hashgrid_query_cuda(coords, codebook, resolution, codebook_bitwidth, probe_bitwidth):
    """hashgrid_query.cpp:41-67: std::vector<Tensor> codebook, std::vector<int32_t> resolution -> [N, 8, L * P * F]."""
    return hashgrid_query(coords, list(codebook), _resolution_list(resolution), int(codebook_bitwidth), int(probe_bitwidth))


def _ref_hashgrid_query_backward_cuda(coords, grad_output, resolution, codebook_shapes, codebook_bitwidth, feature_dim,
                                      probe_bitwidth):
    """hashgrid_query.cpp:69-97 -> std::vector<Tensor> of gradient tables."""
    return hashgrid_query_backward(coords, grad_output, _resolution_list(resolution), list(codebook_shapes),
                                   int(codebook_bitwidth), int(feature_dim), int(probe_bitwidth))


ops = _Namespace("wisp._C.ops",
                 hashgrid_query_cuda=_ref_hashgrid_query_cuda,
                 hashgrid_query_backward_cuda=_ref_hashgrid_query_backward_cuda,
                 grid_interpolate_cuda=_ref_grid_interpolate_cuda,
                 grid_interpolate_backward_cuda=_ref_grid_interpolate_backward_cuda,
                 hashgrid_interpolate_cuda=_ref_hashgrid_interpolate_cuda,
                 hashgrid_interpolate_backward_cuda=_ref_hashgrid_interpolate_backward_cuda,
                 uniform_sample_cuda=_ref_uniform_sample_cuda)
render = _Namespace("wisp._C.render", find_depth_bound_cuda=_ref_find_depth_bound_cuda)
