"""CPU: the C-ABI library loads and exports every symbol include/wisp_hip.h declares; host-side logic of the wisp
mirror (value types, SPC build, channel / kwarg plumbing, constructor schemas) behaves like the reference's."""
import ctypes
import inspect
import os
import pickle
import re
import types

import numpy as np
import pytest
import torch

from oracle import spc as ospc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "wisp_hip.h")).read()
    declared = set(re.findall(r"\b(wisp_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = ctypes.CDLL(os.path.join(ROOT, "kaolin-wisp_amd", "csrc", "libwisp_hip.so"))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    import wisp._C as C
    assert set(C.SIGNATURES) == declared           # the Python binding covers the whole header, nothing else
    assert C.lib.wisp_abi_version() == 4 == C.ABI_VERSION
    assert C.lib.wisp_nerf_mlp_param_count(32, 64, 4) == 3152 + 7107    # decoder sizes of nerf_hash.yaml


def test_ctypes_signatures_follow_the_header_argument_by_argument():
    """Every declaration of include/wisp_hip.h against wisp._C.SIGNATURES: the same number of arguments, pointers bound as
    c_void_p, 64-bit integers as c_int64, ints as c_int32, floats as c_float, the stream as c_void_p - a binding that drifts from
    the header (an argument added on one side only) shifts every later argument silently at call time."""
    import wisp._C as C
    header = open(os.path.join(ROOT, "include", "wisp_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)                       # comments may hold commas and parentheses
    header = re.sub(r"//[^\n]*", " ", header)
    decls = re.findall(r"\b(?:int|int64_t|const char\s*\*|void\s*\*|void)\s+(wisp_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", header)
    assert len(decls) == len(C.SIGNATURES)
    kinds = {ctypes.c_void_p: "ptr", ctypes.c_int64: "i64", ctypes.c_int32: "i32", ctypes.c_float: "f32", ctypes.c_uint64: "u64",
             ctypes.c_uint32: "u32", ctypes.c_double: "f64"}

    def kind_of(arg):
        arg = " ".join(arg.split())
        if "*" in arg or arg.startswith("wisp_stream_t"):
            return "ptr"
        base = arg.rsplit(" ", 1)[0].replace("const ", "")
        return {"int64_t": "i64", "uint64_t": "u64", "int": "i32", "int32_t": "i32", "uint32_t": "u32", "float": "f32", "double": "f64"}[base]

    for name, args in decls:
        args = [a for a in (x.strip() for x in args.split(",")) if a and a != "void"]
        want = [kind_of(a) for a in args]
        got = [kinds[t] for t in C.SIGNATURES[name]]
        assert got == want, (name, got, want)


def test_hot_path_refuses_cpu_tensors():
    import wisp._C as C
    with pytest.raises(RuntimeError, match="GPU tensor"):
        C.hashgrid_interpolate(torch.zeros(4, 3), torch.zeros(8, 2), torch.zeros(2, dtype=torch.int64), [2], 3)
    from wisp.accelstructs import OctreeAS
    blas = OctreeAS.make_dense(2)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        blas.query(torch.zeros(4, 3))


def test_raymarch_coarse_level_choice():
    """Host-side choice of the occupancy level the 'ray' count kernel stages in LDS: the level whose cell is about one
    64-candidate chunk long, never finer than level 5 (LDS budget) nor than marching level - 1."""
    import wisp._C as C
    assert C.raymarch_coarse_level(1.0, 5.0, 2048, 7) == 4          # nerf_hash.yaml: chunk 0.125 = a level-4 cell
    assert C.raymarch_coarse_level(1.0, 5.0, 512, 7) == 2           # chunk 0.5
    assert C.raymarch_coarse_level(1.0, 5.0, 65536, 9) == 5         # capped by the LDS budget
    assert C.raymarch_coarse_level(1.0, 5.0, 2048, 3) == 2          # capped by the marching level
    assert C.raymarch_coarse_level(1.0, 5.0, 64, 7) is None         # one chunk spans the whole scene
    assert C.raymarch_coarse_level(2.0, 2.0, 128, 7) is None        # empty depth range
    assert C.raymarch_coarse_level(0.0, 6.0, 1024, 1) is None       # nothing coarser than level 1 is worth a test


def test_no_product_code_imports_the_oracle():
    pkg = os.path.join(ROOT, "kaolin-wisp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(dirpath, f)


def test_rays_api():
    from wisp.core import Rays
    r = Rays(torch.rand(10, 3), torch.rand(10, 3), dist_min=1.0, dist_max=5.0)
    assert len(r) == 10 and r.shape == (10,) and r.ndim == 1
    a, b = r.split(6)
    assert len(a) == 6 and len(b) == 4 and a.dist_max == 5.0
    c = Rays.cat([a, b])
    assert torch.equal(c.origins, r.origins)
    assert Rays.stack([a[:4], b]).origins.shape == (2, 4, 3)
    assert r.reshape(2, 5, 3).shape == (2, 5) and r[2:4].origins.shape == (2, 3)
    assert r.to(torch.float64).origins.dtype == torch.float64 and r.to(torch.float32) is r
    with pytest.raises(Exception):
        len(Rays(torch.rand(3, 3), torch.rand(4, 3)))


def test_render_buffer_api():
    from wisp.core import RenderBuffer
    a = RenderBuffer(rgb=torch.rand(4, 3), alpha=torch.rand(4, 1), hit=torch.ones(4, dtype=torch.bool))
    b = RenderBuffer(rgb=torch.rand(2, 3), alpha=torch.rand(2, 1), hit=torch.zeros(2, dtype=torch.bool))
    assert a.depth is None and a.nonexistent is None
    assert a.channels == {"rgb", "alpha", "hit"}
    c = a + b
    assert c.rgb.shape == (6, 3) and c.hit.shape == (6,) and c.rgba.shape == (6, 4)
    assert c.reshape(2, 3, -1).rgb.shape == (2, 3, 3)
    assert pickle.loads(pickle.dumps(c)).rgb.shape == (6, 3)
    assert set(dict(iter(c)).keys()) >= {"rgb", "alpha", "depth", "hit"}
    assert c.half().rgb.dtype == torch.float16 and "rgb" in c.numpy_dict()


@pytest.mark.parametrize("level", [1, 3])
def test_dense_spc_build_matches_oracle(level):
    from wisp.ops import spc as wspc
    oc = wspc.create_dense_octree(level).cpu()
    assert np.array_equal(oc.numpy(), ospc.create_dense_octree(level))
    pts, pyr, ex = wspc.octree_to_spc(oc)
    opts, opyr, oex = ospc.octree_to_spc(oc.numpy())
    assert np.array_equal(pts.numpy(), opts) and np.array_equal(pyr.numpy(), opyr) and np.array_equal(ex.numpy(), oex)


def test_sparse_spc_build_dual_and_pointcloud_match_oracle():
    from wisp.ops import spc as wspc
    rng = np.random.default_rng(1)
    P = rng.integers(0, 32, size=(400, 3))
    oc = wspc.unbatched_points_to_octree(torch.from_numpy(P).short(), 5).cpu()
    assert np.array_equal(oc.numpy(), ospc.points_to_octree(P, 5))
    pts, pyr, ex = wspc.octree_to_spc(oc)
    opts, opyr, oex = ospc.octree_to_spc(oc.numpy())
    assert np.array_equal(pts.numpy(), opts) and np.array_equal(pyr.numpy(), opyr)
    pd, pyd, tr, par = wspc.make_trilinear_spc(pts, pyr)
    opd, opyd = ospc.make_dual(opts, opyr)
    otr, opar = ospc.make_trinkets(opts, opyr, opd, opyd)
    assert np.array_equal(pd.numpy(), opd) and np.array_equal(pyd.numpy(), opyd)
    assert np.array_equal(tr.numpy(), otr) and np.array_equal(par.numpy(), opar)
    cloud = rng.uniform(-1, 1, (1000, 3)).astype(np.float32)
    oc2 = wspc.pointcloud_to_octree(torch.from_numpy(cloud), 4).cpu()
    assert np.array_equal(oc2.numpy(), ospc.pointcloud_to_octree(cloud, 4))
    assert np.array_equal(wspc.quantize_points(torch.from_numpy(cloud), 4).numpy(), ospc.quantize_points(cloud, 4))


def test_octree_as_attributes_and_aabb():
    from wisp.accelstructs import OctreeAS, AxisAlignedBBoxAS
    b = OctreeAS.make_dense(3)
    assert b.max_level == 3 and b.occupancy() == [1, 8, 64] and b.capacity() == [1, 8, 64] and b.name() == "Octree"
    assert b.points.dtype == torch.int16 and b.prefix.dtype == torch.int32 and b.octree.dtype == torch.uint8
    a = AxisAlignedBBoxAS()
    assert isinstance(a, OctreeAS) and a.max_level == 1
    with pytest.raises(TypeError):
        from wisp.core import Rays
        b.raymarch(Rays(torch.zeros(1, 3), torch.ones(1, 3)), 'spiral', 4)
    pickle.loads(pickle.dumps(b))


def test_constructor_schemas_match_reference_api():
    """Arg names + defaults are the config schema of the reference."""
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    from wisp.accelstructs import ASRaymarchResults
    sig = inspect.signature(HashGrid.__init__)
    assert list(sig.parameters)[1:] == ['blas', 'feature_dim', 'resolutions', 'multiscale_type', 'feature_std',
                                        'feature_bias', 'codebook_bitwidth', 'coord_dim']
    sig = inspect.signature(HashGrid.from_geometric)
    assert list(sig.parameters) == ['blas', 'feature_dim', 'num_lods', 'multiscale_type', 'feature_std', 'feature_bias',
                                    'codebook_bitwidth', 'min_grid_res', 'max_grid_res', 'coord_dim']
    sig = inspect.signature(NeuralRadianceField.__init__)
    assert list(sig.parameters)[1:] == ['grid', 'pos_embedder', 'view_embedder', 'pos_multires', 'view_multires',
                                        'position_input', 'activation_type', 'layer_type', 'hidden_dim', 'num_layers',
                                        'bias', 'prune_density_decay', 'prune_min_density']
    assert sig.parameters['hidden_dim'].default == 128 and sig.parameters['prune_min_density'].default == 0.6
    sig = inspect.signature(PackedRFTracer.__init__)
    assert [(k, v.default) for k, v in list(sig.parameters.items())[1:]] == [
        ('raymarch_type', 'ray'), ('num_steps', 1024), ('step_size', 1.0), ('bg_color', (1.0, 1.0, 1.0))]
    assert [f for f in ASRaymarchResults.__dataclass_fields__] == ['samples', 'ridx', 'depth_samples', 'deltas',
                                                                    'boundary', 'pack_info']


def test_model_layout_and_state_dict_names():
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    g = HashGrid.from_geometric(OctreeAS.make_dense(2), feature_dim=2, num_lods=16, multiscale_type='cat',
                                feature_std=1e-9, codebook_bitwidth=19, min_grid_res=16, max_grid_res=512)
    assert g.resolutions == [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512]
    assert g.codebook.feats.shape == (5217937, 2)                # SURVEY.md section 8, C2 sizes
    nef = NeuralRadianceField(g, view_embedder='positional', hidden_dim=64, bias=True,
                              prune_density_decay=0.95, prune_min_density=2.956)
    names = [n for n, _ in nef.named_parameters()]
    assert names == ['grid.codebook.feats', 'view_embedder.bands', 'decoder_density.layers.0.weight',
                     'decoder_density.layers.0.bias', 'decoder_density.lout.weight', 'decoder_density.lout.bias',
                     'decoder_color.layers.0.weight', 'decoder_color.layers.0.bias', 'decoder_color.layers.1.weight',
                     'decoder_color.layers.1.bias', 'decoder_color.lout.weight', 'decoder_color.lout.bias']
    assert nef.decoder_density.lout.bias[0].item() == 1.0
    assert nef.decoder_color.layers[0].weight.shape == (64, 42)
    assert nef.get_supported_channels() == {"rgb", "density"}
    with pytest.raises(Exception, match="not supported"):
        nef(channels="sdf", coords=torch.zeros(1, 3))


def test_tracer_fills_trace_kwargs_from_attributes():
    from wisp.tracers import PackedRFTracer
    seen = {}

    class Probe(PackedRFTracer):
        def trace(self, nef, rays, channels, extra_channels, lod_idx=None, raymarch_type='voxel', num_steps=64,
                  step_size=1.0, bg_color='white', jitter=None):
            seen.update(raymarch_type=raymarch_type, num_steps=num_steps, channels=channels, extra=extra_channels)
            return "rb"

    class Nef:
        def get_supported_channels(self):
            return {"rgb", "density"}

    t = Probe(raymarch_type='ray', num_steps=2048)
    assert t(Nef(), rays=None, channels=["rgb"]) == "rb"
    assert seen["raymarch_type"] == 'ray' and seen["num_steps"] == 2048 and seen["channels"] == {"rgb"}   # quirk 6
    t(Nef(), rays=None, channels="rgb", num_steps=7)
    assert seen["num_steps"] == 7
    with pytest.raises(Exception, match="not supported"):
        t(Nef(), rays=None, channels=["normal"])


def test_dataset_batch_and_ray_sampler_contract():
    from wisp.core import Rays
    from wisp.datasets import MultiviewBatch, SampleRays, MultiviewTensorDataset
    V, P = 3, 50
    o, d, rgb = torch.rand(V, P, 3), torch.rand(V, P, 3), torch.rand(V, P, 3)
    tf = SampleRays(num_samples=16)
    ds = MultiviewTensorDataset(o, d, rgb, dist_min=1.0, dist_max=5.0, transform=tf)
    assert len(ds) == V
    item = ds[1]
    assert item['rays'].origins.shape == (16, 3) and item['rgb'].shape == (16, 3) and item['rays'].dist_max == 5.0
    tf.set_num_samples(7)                      # what calc_adaptive_rays does every step
    assert ds[0]['rays'].origins.shape == (7, 3)
    b = MultiviewBatch(rays=Rays(o[0], d[0]), rgb=rgb[0], mask=torch.ones(P, 1))
    assert set(b.fields) == {"rays", "cameras", "rgb", "mask"} and set(b.ray_values()) == {"rgb", "mask"}
    g = torch.Generator().manual_seed(1)
    a1 = SampleRays(8)(b, generator=g)['rgb']
    g = torch.Generator().manual_seed(1)
    a2 = SampleRays(8)(b, generator=g)['rgb']
    assert torch.equal(a1, a2)
    # a row of a picked ray stays aligned with its colour
    idx = (b['rgb'][:, None, :] == a1[None]).all(-1).float().argmax(0)
    assert torch.equal(b['rays'].origins[idx], SampleRays(8)(b, generator=torch.Generator().manual_seed(1))['rays'].origins)


def test_checkpoint_roundtrip_keeps_pruned_octree(tmp_path):
    from wisp.accelstructs import OctreeAS
    from wisp.models import Pipeline
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import save_pipeline, load_pipeline

    def make(points):
        blas = OctreeAS.from_quantized_points(points, 3)
        grid = HashGrid.from_geometric(blas, feature_dim=2, num_lods=4, multiscale_type='cat', feature_std=0.1,
                                       codebook_bitwidth=8, min_grid_res=4, max_grid_res=16)
        return Pipeline(NeuralRadianceField(grid, view_embedder='positional', hidden_dim=16, bias=True), PackedRFTracer())
    pts = torch.tensor([[0, 0, 0], [7, 7, 7], [3, 4, 5]], dtype=torch.int16)
    p1 = make(pts)
    for fmt in ("full", "state_dict"):
        path = str(tmp_path / f"m_{fmt}.pth")
        save_pipeline(p1, path, fmt)
        p2 = load_pipeline(path, pipeline=make(torch.tensor([[1, 1, 1]], dtype=torch.int16)))
        assert torch.equal(p2.nef.grid.blas.octree.cpu(), p1.nef.grid.blas.octree.cpu())
        assert torch.equal(p2.nef.grid.codebook.feats.cpu(), p1.nef.grid.codebook.feats.cpu())
    # 'state_dict' files are the bare OrderedDict the reference writes (base_trainer.py:357) and can load
    import collections
    raw = torch.load(str(tmp_path / "m_state_dict.pth"), weights_only=False)
    assert isinstance(raw, collections.OrderedDict) and "nef.grid.codebook.feats" in raw and "state_dict" not in raw
    p3 = make(torch.tensor([[1, 1, 1]], dtype=torch.int16))
    p3.load_state_dict(raw)                                     # what reference code does with the file
    # a reference-written file (no sidecar) loads too, and the wrapped files of earlier versions are still read
    ref_path = str(tmp_path / "ref_style.pth")
    torch.save(p1.state_dict(), ref_path)
    p4 = load_pipeline(ref_path, pipeline=make(torch.tensor([[1, 1, 1]], dtype=torch.int16)))
    assert torch.equal(p4.nef.grid.codebook.feats, p1.nef.grid.codebook.feats)
    old_path = str(tmp_path / "old_style.pth")
    torch.save({"state_dict": p1.state_dict(), "blas_octree": p1.nef.grid.blas.octree, "grid_occupancy": None}, old_path)
    p5 = load_pipeline(old_path, pipeline=make(torch.tensor([[1, 1, 1]], dtype=torch.int16)))
    assert torch.equal(p5.nef.grid.blas.octree.cpu(), p1.nef.grid.blas.octree.cpu())


def test_table_aux_is_validated_and_never_pickled():
    """the bf16 shadow / flat gradient buffer of a table Parameter must not be trusted blindly nor pickled."""
    import pickle
    from wisp.ops import grid as G
    p = torch.nn.Parameter(torch.randn(64, 2))
    p.grad = torch.zeros(64, 2)
    shadow = p.detach().to(torch.bfloat16)
    G.register_table_aux(p, shadow=shadow, grad_buffer=p.grad)
    assert G.current_shadow(p, torch.bfloat16) is shadow and G.current_grad_buffer(p) is p.grad
    assert G.current_shadow(p, torch.float16) is None
    with torch.no_grad():
        p.copy_(torch.randn(64, 2))                             # e.g. load_state_dict: bumps the version counter
    assert G.current_shadow(p, torch.bfloat16) is None          # stale copy refused
    G.mark_shadow_current(p)                                    # what the fused optimizer step announces
    assert G.current_shadow(p, torch.bfloat16) is shadow
    p.grad = torch.zeros(64, 2)                                 # a different storage: registered buffer no longer aliases .grad
    assert G.current_grad_buffer(p) is None
    assert "_wisp" not in repr(sorted(p.__dict__)) and b"_wisp" not in pickle.dumps(p)


def test_trainer_prune_is_a_noop_without_an_occupancy_record():
    """prune_every=100 with an Octree/Codebook/Triplanar grid must do nothing (reference: nerf.py:181-183)."""
    from wisp.trainers.multiview_trainer import MultiviewTrainStep

    class Grid:
        pass

    class Nef:
        prune_density_decay, prune_min_density, grid = 0.95, 2.0, Grid()

        def prune(self, **kw):
            raise AssertionError("must not be reached")

    class Pipe:
        nef = Nef()
    t = MultiviewTrainStep.__new__(MultiviewTrainStep)
    t.pipeline, t._params_ready = Pipe(), None
    t.prune()
    Nef.prune_density_decay = None
    Nef.grid.dense_points = torch.zeros(4, 3)
    t.prune()


def test_hashgrid_naive_agrees_with_oracle_on_cpu():
    from wisp.ops.grid import hashgrid_naive
    from oracle import hashgrid as oh
    res, bw = [4, 8, 16, 40], 10
    sizes, begin = oh.table_layout(res, 2 ** bw)
    torch.manual_seed(0)
    table = torch.randn(int(begin[-1]), 2)
    coords = torch.rand(500, 3) * 2 - 1
    got = hashgrid_naive(coords, torch.tensor(res), bw, 3, table, sizes, begin[:-1])
    want = oh.hashgrid_forward(coords, table, torch.from_numpy(begin), res, bw)
    assert got.shape == (500, 8) and float((got - want).abs().max()) < 1e-5
    assert hashgrid_naive(coords, res, bw, 1, table, sizes, begin[:-1]).shape == (500, 4)


def test_reference_named_native_surface_exists():
    """wisp._C.ops.* / wisp._C.render.* under the names and arities the reference binds (wisp/csrc/bindings.cpp:21-35)."""
    import inspect
    import wisp._C as C
    want = {"ops": {"hashgrid_interpolate_cuda": 5, "hashgrid_interpolate_backward_cuda": 8, "uniform_sample_cuda": 4,
                    "grid_interpolate_cuda": 2, "grid_interpolate_backward_cuda": 3,
                    "hashgrid_query_cuda": 5, "hashgrid_query_backward_cuda": 7},
            "render": {"find_depth_bound_cuda": 3}}
    for ns, fns in want.items():
        for name, arity in fns.items():
            fn = getattr(getattr(C, ns), name)
            assert len(inspect.signature(fn).parameters) == arity, name


@pytest.mark.skipif(not os.path.isfile("/root/reference/wisp/ops/grid.py"), reason="reference tree not mounted (GPU box)")
def test_reference_ops_grid_module_binds_to_this_C_unchanged():
    """Execute the reference's OWN wisp/ops/grid.py source (read in place, never copied) with `wisp._C` = this package's
    module and a stub for the absent kaolin import: every `wisp_C.<ns>.<fn>` it names must resolve here."""
    import re
    import sys
    import types
    import wisp._C as C
    src = open("/root/reference/wisp/ops/grid.py").read()
    used = set(re.findall(r"wisp_C\.(\w+)\.(\w+)", src))
    assert ("ops", "hashgrid_interpolate_cuda") in used and ("ops", "hashgrid_interpolate_backward_cuda") in used
    for ns, fn in used:                                   # every native function the reference's grid.py names
        assert callable(getattr(getattr(C, ns), fn)), (ns, fn)
    assert {fn for _, fn in used} >= {"hashgrid_query_cuda", "hashgrid_query_backward_cuda", "grid_interpolate_cuda"}
    stub = types.ModuleType("kaolin"); stub.ops = types.ModuleType("kaolin.ops"); stub.ops.spc = types.ModuleType("kaolin.ops.spc")
    stub._C = types.ModuleType("kaolin._C")
    saved = {k: sys.modules.get(k) for k in ("kaolin", "kaolin.ops", "kaolin.ops.spc")}
    sys.modules.update({"kaolin": stub, "kaolin.ops": stub.ops, "kaolin.ops.spc": stub.ops.spc})
    try:
        ns = {"__name__": "reference_ops_grid"}
        exec(compile(src, "/root/reference/wisp/ops/grid.py", "exec"), ns)      # `import wisp._C as wisp_C` -> this package
        assert ns["wisp_C"] is C and issubclass(ns["HashGridInterpolate"], torch.autograd.Function)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_allreduce_range_leaves_out_only_a_gradient_free_tail():
    """MultiviewTrainStep._live_grad_numel (host logic): under 'cat' with lod_idx = num_lods - 1 the finest level's rows get no
    gradient; they are skipped by the all-reduce only when they really are the tail of the flat buffer (one table, last
    tensor, nothing after it) and the step went through the direct path."""
    import types
    from wisp.trainers import FlatParams, MultiviewTrainStep

    class Field(torch.nn.Module):
        def __init__(self, with_rest):
            super().__init__()
            self.decoder = torch.nn.Linear(4, 3)
            self.grid_table = torch.nn.Parameter(torch.zeros(10 + 20 + 30, 2))          # three levels: 10, 20, 30 rows
            if with_rest:
                self.extra = torch.nn.Parameter(torch.zeros(5))

    def trainer(with_rest):
        m = Field(with_rest)
        t = MultiviewTrainStep.__new__(MultiviewTrainStep)
        t.flat = FlatParams(m)
        t._direct = types.SimpleNamespace(hash_fast=True, table=m.grid_table, zero_from_col=2 * 2, res=[4, 8, 16],
                                          first_idx=torch.tensor([0, 10, 30, 60]))
        t._last_step_modular = False
        return t, m

    t, m = trainer(False)
    table_off = (m.grid_table.data_ptr() - t.flat.data.data_ptr()) // 4
    assert t._live_grad_numel() == table_off + 30 * 2                                   # levels 0 and 1 travel, level 2 does not
    assert t._live_grad_numel() < t.flat.grad.numel()
    t._direct.zero_from_col = 3 * 2                                                     # every level has a gradient
    assert t._live_grad_numel() == t.flat.grad.numel()
    t._direct.zero_from_col = 2 * 2
    t._last_step_modular = True                                                         # autograd path: no assumption
    assert t._live_grad_numel() == t.flat.grad.numel()
    t2, _ = trainer(True)                                                               # a tensor after the table
    assert t2._live_grad_numel() == t2.flat.grad.numel()
    t3, _ = trainer(False)
    t3._direct.hash_fast = False                                                        # another grid's own backward: no assumption
    assert t3._live_grad_numel() == t3.flat.grad.numel()


def test_folded_grid_optimizer_host_logic():
    """MultiviewTrainStep._fused_update_args / _uncovered_grid_ranges (host logic of the AdamW step folded into the hash-grid
    backward): the folded update is offered only where it is the optimizer's own step for the table - one rank, AdamW, the
    two-feature hash tier, nobody having replaced optimizer_step - with the learning rate and step count of the step that is
    about to run; and the ranges left to the separate pass are exactly the complement of what the launch reported."""
    import types
    from wisp.trainers import FlatParams, MultiviewTrainStep

    class Field(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.decoder = torch.nn.Linear(4, 3)
            self.grid_table = torch.nn.Parameter(torch.zeros(10 + 20 + 30 + 40, 2))      # four levels: 10, 20, 30, 40 rows
            self.grid_other = torch.nn.Parameter(torch.zeros(6))                         # something else in the grid group

    m = Field()
    t = MultiviewTrainStep.__new__(MultiviewTrainStep)
    t.flat = FlatParams(m)
    first = [0, 10, 30, 60, 100]
    t._direct = types.SimpleNamespace(hash_fast=True, table=m.grid_table, _first_idx_host=first)
    t.fuse_grid_optimizer, t.optimizer, t.world, t.force_allreduce = True, 'adamw', 1, False
    t.lr, t.grid_lr_weight, t.betas, t.eps, t.weight_decay = 1e-3, 100.0, (0.9, 0.999), 1e-15, 1e-6
    t.milestones, t.gamma, t.opt_steps = [3], 0.5, 1
    # (the flat buffers live on the CPU here: the one condition that cannot be met without a GPU is stubbed)
    class CudaLike(torch.Tensor):
        is_cuda = True
    t.flat.data = t.flat.data.as_subclass(CudaLike)
    a = t._fused_update_args()
    off = (m.grid_table.data_ptr() - t.flat.data.data_ptr()) // 4
    assert a is not None and a["step"] == 2 and abs(a["lr"] - 0.1) < 1e-12 and a["shadow"] is None and a["grad_scale"] == 1.0
    assert a["param"].data_ptr() == m.grid_table.data_ptr() and a["exp_avg"].shape == m.grid_table.shape
    assert a["exp_avg"].data_ptr() == t.flat.exp_avg.data_ptr() + 4 * off
    t.opt_steps = 2                                           # the step about to run is the milestone step
    assert abs(t._fused_update_args()["lr"] - 0.05) < 1e-12
    for attr, bad in (("world", 2), ("optimizer", "rmsprop"), ("force_allreduce", True), ("fuse_grid_optimizer", False)):
        good = getattr(t, attr)
        setattr(t, attr, bad)
        assert t._fused_update_args() is None, attr
        setattr(t, attr, good)
    t.optimizer_step = lambda *a, **k: None                    # a replaced optimizer step gets the whole gradient
    assert t._fused_update_args() is None
    del t.optimizer_step
    assert t._fused_update_args() is not None
    # complement of the covered rows inside the grid group (levels 1 and 3 covered, level 3 only partly)
    ga, gb = t.flat.ranges["grid"]
    left = t._uncovered_grid_ranges([0, 20, 0, 25])
    assert left == [(ga, off + 10 * 2), (off + 30 * 2, off + 60 * 2), (off + 85 * 2, gb)]
    assert t.fused_elements_last == (20 + 25) * 2
    assert t._uncovered_grid_ranges([0, 0, 0, 0]) == [(ga, gb)] and t.fused_elements_last == 0
    # a count beyond the level's rows is clipped (the library reports its plan, the table may own fewer rows)
    assert t._uncovered_grid_ranges([99, 0, 0, 0])[0][0] == off + 10 * 2


def test_look_ahead_raytrace_state_is_only_used_where_it_fits(monkeypatch):
    """OctreeAS.raytrace(..., begun=state) / raymarch(..., begin_only / begun): host plumbing of the one-batch look-ahead with the HIP
    calls replaced by recorders - a state issued for the same Rays object, level and octree is finished as it is; one issued for
    other rays, another level or another OctreeAS (a prune replaces the object) is dropped and the count redone; begin_only
    returns the state for 'voxel' / 'uniform' and None for 'ray'; the voxel march hands the nuggets on as cell hints."""
    import types
    import wisp.accelstructs.octree_as as mod
    from wisp.accelstructs import OctreeAS
    from wisp.core import Rays
    calls = []

    def begin(octree, points, exsum, origins, dirs, level):
        calls.append(("begin", level))
        return dict(level=level, offsets=torch.zeros(origins.shape[0] + 1, dtype=torch.int64), tag=len(calls))

    def finish(st, with_exit=False):
        calls.append(("finish", st["tag"], with_exit))
        n = 3
        return (torch.zeros(n, dtype=torch.int32), torch.arange(n, dtype=torch.int32), torch.zeros(n, 2 if with_exit else 1), st["offsets"])

    def voxel(origins, dirs, ridx, depth, num_samples, jitter, seed):
        S = ridx.shape[0] * num_samples
        return (torch.zeros(S, dtype=torch.int64), torch.zeros(S, 3), torch.zeros(S, 1), torch.zeros(S, 1), torch.zeros(S, dtype=torch.bool))

    fake = types.SimpleNamespace(spc_raytrace_begin=begin, spc_raytrace_finish=finish, raymarch_voxel=voxel)
    monkeypatch.setattr(mod, "_hip", lambda: fake)
    oc, pts, pyr, ex = (torch.from_numpy(np.asarray(a)) for a in (lambda o: (o,) + ospc.octree_to_spc(o))(ospc.create_dense_octree(2)))
    oc._wisp_spc_parts = (pts.short(), pyr.int(), ex.int())
    blas = OctreeAS(oc)
    monkeypatch.setattr(blas, "_to_device", lambda dev: None)
    r0 = Rays(torch.zeros(4, 3), torch.ones(4, 3), dist_min=0.0, dist_max=1.0)
    r1 = Rays(torch.zeros(4, 3), torch.ones(4, 3), dist_min=0.0, dist_max=1.0)
    st = blas.raytrace_begin(r0, 2)
    assert st["blas"] is blas and st["rays"] is r0 and calls == [("begin", 2)]
    blas.raytrace(r0, 2, with_exit=True, begun=st)
    assert calls[-1] == ("finish", 1, True) and len(calls) == 2                       # finished as issued, nothing recounted
    for rays, level, owner in ((r1, 2, blas), (r0, 1, blas), (r0, 2, None)):
        stale = blas.raytrace_begin(r0, 2)
        if owner is None:
            stale["blas"] = object()                                                   # an octree a prune has replaced
        before = len(calls)
        blas.raytrace(rays, level, begun=stale)
        assert [c[0] for c in calls[before:]] == ["begin", "finish"] and calls[-1][1] != stale["tag"]
    assert blas.raymarch(r0, 'ray', 8, level=2, begin_only=True) is None
    st = blas.raymarch(r0, 'voxel', 4, level=2, begin_only=True)
    assert st["level"] == 2 and st["rays"] is r0
    before = len(calls)
    monkeypatch.setattr(OctreeAS, "_draw_seed", staticmethod(lambda: 7))
    rm = blas.raymarch(r0, 'voxel', 4, level=2, begun=st)
    assert [c[0] for c in calls[before:]] == ["finish"] and rm.samples.shape[0] == 12
    assert rm.nugget_level == 2 and rm.samples_per_nugget == 4 and rm.nugget_pidx.shape[0] == 3


def test_hashgrid_backward_workspace_query_is_sane_without_a_gpu():
    """wisp_hashgrid_bwd_workspace_bytes is pure host arithmetic (slot plan of the binned backward, incl. the capped grid of
    the queue emitter): callable without a device, monotone in the sample count, and a few GB at the nerf_hash shape."""
    import ctypes
    import wisp._C as C
    res = [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512]
    arr = (ctypes.c_int32 * len(res))(*res)
    sizes = [int(C.lib.wisp_hashgrid_bwd_workspace_bytes(n, 3, -1, 2, arr, len(res), 19, None)) for n in (4096, 1 << 16, 1 << 18, 1 << 21)]
    assert all(b > 0 for b in sizes) and sizes == sorted(sizes)
    assert (1 << 30) < sizes[-1] < (6 << 30)
    assert int(C.lib.wisp_hashgrid_bwd_workspace_bytes(0, 3, -1, 2, arr, len(res), 19, None)) == 0
    assert int(C.lib.wisp_hashgrid_bwd_workspace_bytes(1 << 18, 4, -1, 2, arr, len(res), 19, None)) == 0       # bad coord_dim
    # per-level slot scales (what wisp._C._SlotFit learns from the launches: the run merge leaves ~1/20 of the no-merge record
    # count on the coarsest level and ~1/2 on the finest): the scratch follows them - VERDICT r2 #9 asks for <= 2 x what is written,
    # 0.43 GB at this shape - and is monotone in every scale; scales of 1 are the unscaled plan
    def scaled(v):
        return int(C.lib.wisp_hashgrid_bwd_workspace_bytes(1 << 21, 3, C.BF16, 2, arr, len(res), 19, ctypes.cast((ctypes.c_float * 16)(*v), ctypes.c_void_p)))
    ones = scaled([1.0] * 16)                                                                          # bf16: 8-byte records, capped grid
    assert (2 << 30) < ones <= sizes[-1]
    measured_like = [0.05, 0.05, 0.06, 0.07, 0.08, 0.1, 0.12, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.5, 0.6, 0.7]
    fitted = scaled([1.35 * f for f in measured_like])
    assert fitted < 0.5 * ones and fitted < (1 << 30)
    assert scaled([0.5] * 16) < ones and scaled([0.25] * 16) < scaled([0.5] * 16)
    assert scaled([0.0] * 16) == ones and scaled([float('nan')] * 16) == ones                          # nonsense scales are ignored


def test_octree_from_mesh_covers_the_surface(tmp_path):
    """OctreeAS.from_mesh (octree_as.py:65-106) on a quad-faced cube OBJ: after sphere normalisation the cube has half-side
    1/sqrt(3); every occupied level-4 cell must touch its surface, and every cell the six faces pass through must be occupied
    (5e5 samples on 6 faces leave no holes at 16^3)."""
    from wisp.accelstructs import OctreeAS
    from wisp.ops import mesh as mesh_ops
    obj = tmp_path / "cube.obj"
    v = [(x, y, z) for x in (0, 2) for y in (0, 2) for z in (0, 2)]
    quads = [(1, 2, 4, 3), (5, 7, 8, 6), (1, 5, 6, 2), (3, 4, 8, 7), (1, 3, 7, 5), (2, 6, 8, 4)]
    obj.write_text("# cube\n" + "".join(f"v {a} {b} {c}\n" for a, b, c in v) + "".join("f " + " ".join(f"{i}/{i}" for i in q) + "\n" for q in quads))
    V, F = mesh_ops.load_obj(str(obj))
    assert V.shape == (8, 3) and F.shape == (12, 3)
    Vn, _ = mesh_ops.normalize(V, F, 'sphere')
    assert abs(float(Vn.norm(dim=1).max()) - 1.0) < 1e-6
    pts, nrm = mesh_ops.sample_surface(Vn, F, 2000)
    h = 1.0 / 3 ** 0.5
    assert float((pts.abs().max(dim=1)[0] - h).abs().max()) < 1e-5 and nrm.shape == (2000, 3)
    torch.manual_seed(0)
    blas = OctreeAS.from_mesh(str(obj), level=4, num_samples_on_mesh=500000)
    cells = blas.points[int(blas.pyramid[1, 4]):].float()                 # level-4 cell coordinates
    lo, hi = cells / 16 * 2 - 1, (cells + 1) / 16 * 2 - 1
    half = 1.0 / 32                                                      # jitter reach of the augmented samples
    # distance of a cell (box) to the cube surface: the box must straddle |x|_inf = h within the jitter reach
    far = torch.maximum(lo.abs(), hi.abs()).max(dim=1)[0]
    near = torch.where((lo <= 0) & (hi >= 0), torch.zeros_like(lo), torch.minimum(lo.abs(), hi.abs())).max(dim=1)[0]
    assert bool(((near <= h + half) & (far >= h - half)).all())
    occ = torch.zeros(16, 16, 16, dtype=torch.bool)
    occ[cells[:, 0].long(), cells[:, 1].long(), cells[:, 2].long()] = True
    g = torch.linspace(-h, h, 97)
    a, b = torch.meshgrid(g, g, indexing='ij')
    for axis in range(3):
        for s in (-h, h):
            p = torch.stack([torch.full_like(a, s) if k == axis else (a if k == (axis + 1) % 3 else b) for k in range(3)], -1).reshape(-1, 3)
            q = torch.clamp(torch.floor(16 * (0.5 * p + 0.5)), 0, 15).long()
            assert bool(occ[q[:, 0], q[:, 1], q[:, 2]].all())
    assert blas.extent['vertices'].shape == (8, 3) and blas.max_level == 4
    with pytest.raises(NotImplementedError):
        OctreeAS.from_mesh(str(obj), level=3, sample_tex=True)


def test_sharded_optimizer_state_dict_guard_is_weak_and_pickles_inert():
    """The state_dict pre-hook of the opt-in sharded optimizer refuses a checkpoint while other ranks' table slices are stale, does
    not keep the trainer alive, and a module carrying it still pickles (torch.save(pipeline), the reference's 'full' format) - the
    copy that comes back carries an inert guard."""
    import gc
    import io
    import pickle
    import weakref
    from wisp.trainers.multiview_trainer import _StaleMasterGuard

    class Trainer:
        _master_stale = True

    t = Trainer()
    mod = torch.nn.Linear(2, 2)
    mod.register_state_dict_pre_hook(_StaleMasterGuard(t))
    with pytest.raises(RuntimeError, match="sync_master"):
        mod.state_dict()
    t._master_stale = False
    assert len(mod.state_dict()) == 2
    t._master_stale = True
    buf = io.BytesIO()
    torch.save(mod, buf)                                           # pickles the hook table with the module
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert len(back.state_dict()) == 2 and b"Trainer" not in pickle.dumps(_StaleMasterGuard(t))
    alive = weakref.ref(t)
    del t
    gc.collect()
    assert alive() is None and len(mod.state_dict()) == 2         # trainer gone: the guard steps aside


def test_bench_roofline_work_equals_surveys_algorithmic_figures():
    """bench.py's `roofline.achieved` is algorithmic work per unit x units per launch / launch time: the per-sample figures must be
    SURVEY.md 8(d)'s - hash_fwd = 12 + L*2^d*F*b + L*F*b (588 B at L=16, F=2, d=3, 16-bit), hash_bwd = 12 + L*F*b + 2*L*2^d*F*b
    (1100 B), the decoder 20 096 FLOP forward and 3 x that backward - and the peaks the guide's (8 TB/s HBM; 2.5 PFLOP/s dense bf16).
    With the finest level zeroed by the tracer's lod_idx (hash_grid.py:226-229) only the 15 levels that are work done are charged:
    556 / 1036 B."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)                                  # imports only; main() is guarded
    L, F, corners = 16, 2, 8
    for amp, b in ((True, 2), (False, 4)):
        w = bench.work_table(amp, 64)
        assert w["hashgrid_fwd"] == ("hbm", 12 + L * corners * F * b + L * F * b)
        assert w["hashgrid_bwd"] == ("hbm", 12 + L * F * b + 2 * L * corners * F * b)
        assert w["nerf_mlp_fwd"] == ("mfma", 20096) and w["nerf_mlp_bwd"] == ("mfma", 3 * 20096)
        live = bench.work_table(amp, 64, L, L - 1)
        assert live["hashgrid_fwd"] == ("hbm", 12 + (L - 1) * corners * F * b + L * F * b)
        assert live["hashgrid_bwd"] == ("hbm", 12 + L * F * b + 2 * (L - 1) * corners * F * b)
        assert live["nerf_mlp_fwd"] == w["nerf_mlp_fwd"] and live["nerf_mlp_bwd"] == w["nerf_mlp_bwd"]
    assert bench.work_table(True, 64)["hashgrid_fwd"][1] == 588 and bench.work_table(True, 64)["hashgrid_bwd"][1] == 1100
    assert bench.work_table(True, 64, 16, 15)["hashgrid_fwd"][1] == 556 and bench.work_table(True, 64, 16, 15)["hashgrid_bwd"][1] == 1036
    assert bench.work_table(True, 64, 16, 99) == bench.work_table(True, 64)
    assert bench.work_table(True, 128)["nerf_mlp_fwd"][1] == 2 * (32 * 128 + 16 * 128 + 42 * 128 + 128 * 128 + 3 * 128)
    assert bench.HBM_PEAK_GBS == 8000.0
    assert set(bench.PMC_KERNELS) == set(bench.work_table(True))   # every rooflined kernel has its PMC kernel-name list
    # the headline fraction is on the backward's bytes alone: the folded optimizer's bytes have their own key
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"with_fused_optimizer"' in src and "fused_opt_bytes if name" not in src


def test_sdf_fused_field_decision_is_rechecked_when_the_trainer_changes():
    """SDFTrainStep._fused_field cached its yes/no for good at the first step.  The cheap conditions - only_last, the
    field's grid / decoder objects, level count, gradients present - are re-validated on every call now."""
    from wisp.trainers.sdf_trainer import SDFTrainStep
    t = SDFTrainStep.__new__(SDFTrainStep)
    t.only_last = True
    w = [torch.nn.Parameter(torch.zeros(4, 2)) for _ in range(6)]
    for q in w:
        q.grad = torch.zeros_like(q)
    grid = types.SimpleNamespace(features=w[:2], num_lods=2)
    dec = types.SimpleNamespace(layers=[types.SimpleNamespace(weight=w[2], bias=w[3])], lout=types.SimpleNamespace(weight=w[4], bias=w[5]))
    t.nef = types.SimpleNamespace(grid=grid, decoder=dec)
    c = dict(grid=grid, dec=dec, lods=2, prm=list(w))
    t._fused_cache, t._fused_seen_only_last = c, True
    assert t._fused_still_valid(c) and t._fused_field() is c
    w[3].grad = None                                             # zero_grad(set_to_none=True)
    assert not t._fused_still_valid(c)
    w[3].grad = torch.zeros_like(w[3])
    t.nef.grid = types.SimpleNamespace(features=w[:2], num_lods=2)   # another grid object
    assert not t._fused_still_valid(c)
    t.nef.grid = grid
    grid.num_lods = 1
    assert not t._fused_still_valid(c)
    grid.num_lods = 2
    t.only_last = False                                          # the loss now covers every LOD: the fused (finest-LOD) path must step aside
    assert t._fused_field() is None and t._fused_cache is False
    t.only_last = True                                           # ... and is reconsidered when it is switched back (no GPU here: not fusable)
    assert t._fused_cache is False and t._fused_field() is None and t._fused_seen_only_last is True


def test_scratch_buffers_handed_to_a_graph_capture_are_never_replaced_under_it(monkeypatch):
    """per-(device, stream) scratch that was handed out during a capture stays alive and in place; a bigger eager
    request gets a new buffer; a failed call drops the zero-on-return caches."""
    import wisp._C as C
    s = C._Scratch(zeroed=True)
    monkeypatch.setattr(C, "_stream", lambda: types.SimpleNamespace(value=7))
    capturing = [True]
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: capturing[0])
    dev = torch.device("cpu")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    a = s.get(dev, 100)
    assert a.numel() == 125 and bool((a == 0).all()) and s.captured == [a]
    capturing[0] = False
    assert s.get(dev, 120) is a                                   # fits: same buffer
    b = s.get(dev, 1000)                                          # grows: a NEW buffer for eager use ...
    assert b is not a and s.captured == [a] and s.live[(0, 7)] is b          # ... the captured one is still referenced
    C._ZEROED_SCRATCH.append(s)
    try:
        with pytest.raises(RuntimeError):
            C._check(-1, "some_kernel")
    finally:
        C._ZEROED_SCRATCH.remove(s)
    assert s.live == {} and s.captured == [a]


def test_regime_stats_cuts_a_kernel_trace_into_one_table_per_bench_regime(tmp_path):
    """scripts/regime_stats.py (VERDICT r4 weak-2 ii): bench.py brackets every regime with float64 fill launches of
    SENTINEL_ELEMS x (tag + 1) elements; the script pairs them up by grid size and writes one rocprofv3-style kernel-stats table per
    regime, from which `roofline.frac` follows without the other regimes' launches averaged in.  scripts/trace_gaps.py: busy time is the
    UNION of kernel intervals (a copy that overlaps the hash-grid forward is not idle time)."""
    import csv
    import subprocess
    import sys
    spec = __import__("importlib.util").util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = __import__("importlib.util").util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.SENTINEL_ELEMS == 1_000_003 and bench.REGIME_TAGS == {"headline": 0, "large_batch_regime": 1, "dropin_regime": 2}
    fill = ("void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<double>, std::array<char*, 1ul>>"
            "(int, at::native::FillFunctor<double>, std::array<char*, 1ul>)")
    rows, t = [], [0]

    def k(name, dur, grid=64):
        rows.append(dict(Kernel_Name=name, Start_Timestamp=t[0], End_Timestamp=t[0] + dur, Grid_Size_X=grid, Workgroup_Size_X=256))
        t[0] += dur + 10
    k("pretrain_kernel", 999)
    k(fill, 10, grid=977 * 256)                       # tag 0 opens
    for _ in range(5):
        k("hashgrid_bwd_emit_q_kernel", 240); k("hashgrid_bwd_reduce_kernel", 160)
    k(fill, 10, grid=977 * 256)                       # tag 0 closes
    k(fill, 10, grid=2 * 977 * 256)                   # tag 1 opens
    for _ in range(4):
        k("hashgrid_bwd_emit_q_kernel", 60)
    k(fill, 10, grid=2 * 977 * 256)
    d = tmp_path / "prof"
    d.mkdir()
    with open(d / "x_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader(); w.writerows(rows)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "regime_stats.py"), str(d), str(tmp_path / "out")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    head = list(csv.DictReader(open(tmp_path / "out_headline_kernel_stats.csv")))
    assert {h["Name"]: (int(h["Calls"]), float(h["AverageNs"])) for h in head} == {"hashgrid_bwd_emit_q_kernel": (5, 240.0),
                                                                                  "hashgrid_bwd_reduce_kernel": (5, 160.0)}
    ref = list(csv.DictReader(open(tmp_path / "out_large_batch_regime_kernel_stats.csv")))
    assert [(h["Name"], int(h["Calls"]), float(h["AverageNs"])) for h in ref] == [("hashgrid_bwd_emit_q_kernel", 4, 60.0)]
    assert not os.path.exists(tmp_path / "out_dropin_regime_kernel_stats.csv")
    # trace_gaps: a copy overlapping the forward is busy time, not a gap
    rows2, t0 = [], 0
    for step in range(8):
        rows2 += [dict(Kernel_Name="hashgrid_fwd_kernel", Start_Timestamp=t0, End_Timestamp=t0 + 130000),
                  dict(Kernel_Name="copy", Start_Timestamp=t0 + 500, End_Timestamp=t0 + 30000),
                  dict(Kernel_Name="mlp", Start_Timestamp=t0 + 135000, End_Timestamp=t0 + 190000),
                  dict(Kernel_Name="bwd", Start_Timestamp=t0 + 192000, End_Timestamp=t0 + 600000)]
        t0 += 610000
    rows2.append(dict(Kernel_Name="hashgrid_fwd_kernel", Start_Timestamp=t0, End_Timestamp=t0 + 130000))
    d2 = tmp_path / "prof2"
    d2.mkdir()
    with open(d2 / "y_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows2[0]))
        w.writeheader(); w.writerows(rows2)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "trace_gaps.py"), str(d2)], capture_output=True, text=True)
    assert r2.returncode == 0 and "wall 610.0 us  busy 593.0 us  idle 17.0 us" in r2.stdout, r2.stdout[:400]


def test_host_threads_are_bound_to_the_gpus_numa_node(monkeypatch):
    """wisp._C.bind_host_near_device: the sysfs cpulist of the GPU's PCI function becomes this process's affinity (one process per
    GPU on a two-socket node); nothing happens without a topology, with one node, or with WISP_NUMA_BIND=0."""
    import builtins
    import io
    import wisp._C as C
    assert C._parse_cpulist("64-127,192-255\n") == set(range(64, 128)) | set(range(192, 256))
    assert C._parse_cpulist("3") == {3} and C._parse_cpulist("") == set()
    assert C.bind_host_near_device(0) is None                      # no GPU here: nothing to bind to
    props = types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0xf4, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: props)
    files = {"/sys/bus/pci/devices/0000:f4:00.0/numa_node": "1\n", "/sys/bus/pci/devices/0000:f4:00.0/local_cpulist": "2-3,6\n"}
    real_open = builtins.open
    monkeypatch.setattr(builtins, "open", lambda path, *a, **k: io.StringIO(files[path]) if path in files else real_open(path, *a, **k))
    bound = []
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: bound.append(set(cpus)))
    assert C.bind_host_near_device(0) == {"numa_node": 1, "cpus": 3} and bound == [{2, 3, 6}]
    monkeypatch.setenv("WISP_NUMA_BIND", "0")
    assert C.bind_host_near_device(0) is None and len(bound) == 1
    monkeypatch.delenv("WISP_NUMA_BIND")
    files["/sys/bus/pci/devices/0000:f4:00.0/numa_node"] = "-1\n"           # a VM without topology
    assert C.bind_host_near_device(0) is None and len(bound) == 1
    files["/sys/bus/pci/devices/0000:f4:00.0/numa_node"] = "0\n"
    files["/sys/bus/pci/devices/0000:f4:00.0/local_cpulist"] = "0-7\n"       # one node: the affinity already is the list
    assert C.bind_host_near_device(0) is None and len(bound) == 1


def test_strip_dealt_buckets_are_a_bijection_with_exact_reciprocal_division():
    """csrc/hashgrid.hip bucket_of / bucket_row (dense levels of the binned hash-grid backward deal strips of 32 rows to their
    buckets): restated in numpy with the kernel's arithmetic - round = umulhi(strip, ceil(2^32 / buckets)) must equal strip //
    buckets for every strip the plan admits (< 2^18) and every bucket count (<= 1024), every row must land in a bucket below the
    count at an entry below the bucket's 8192 (or 16384 / F) entries, and bucket_row must invert it."""
    rng = np.random.default_rng(5)
    for buckets in [2, 3, 5, 7, 16, 32, 63, 64, 100, 511, 1000, 1024]:
        magic = ((1 << 32) + buckets - 1) // buckets
        strips = np.concatenate([np.arange(0, min(1 << 18, 70000)), rng.integers(0, 1 << 18, 200000), [(1 << 18) - 1]]).astype(np.uint64)
        assert np.array_equal((strips * np.uint64(magic)) >> np.uint64(32), strips // np.uint64(buckets)), buckets
    for csize, entries in [(8192, 25 ** 3), (8192, 32 ** 3), (8192, 80 ** 3), (8192, 50 ** 3), (4096, 20 ** 3), (8192, 8193), (8192, 2 * 8192),
                           (2048, 101 ** 2), (8192, (1 << 23) - 77)]:
        buckets = (entries + csize - 1) // csize
        assert 2 <= buckets <= 1024
        magic = ((1 << 32) + buckets - 1) // buckets
        pw = 1
        while pw * 2 <= buckets:
            pw *= 2
        rot_mask = pw - 1
        row = np.arange(entries, dtype=np.int64)
        strip = row >> 5
        rnd = (strip * magic) >> 32
        t = strip - rnd * buckets + (rnd & rot_mask)
        b = np.where(t >= buckets, t - buckets, t)
        loc = (rnd << 5) | (row & 31)
        assert b.min() >= 0 and b.max() < buckets and loc.max() < csize, (csize, entries)
        # the inverse (bucket_row)
        r2 = loc >> 5
        rot = r2 & rot_mask
        r = np.where(b >= rot, b - rot, b + buckets - rot)
        assert np.array_equal(((r2 * buckets + r) << 5) | (loc & 31), row), (csize, entries)
        # no two rows share (bucket, entry); the buckets' loads differ by at most one strip round
        assert np.unique(b * csize + loc).size == entries
        load = np.bincount(b, minlength=buckets)
        assert load.max() - load.min() <= 64, (csize, entries, load.max(), load.min())


def test_wide_dw2_roles_issue_the_same_barrier_sequence():
    """the producer and consumer waves of wide_dw2_kernel run different functions whose barriers pair up by count
    only.  Every barrier of the two loops is a named WD_STAGE_BARRIER(stage, phase): the two sequences must be identical - stages
    5 .. 1, FILLED then DRAINED - neither loop may hold a bare __syncthreads(), and the kernel gives each role exactly one
    prologue barrier."""
    import re
    src = open(os.path.join(ROOT, "kaolin-wisp_amd", "csrc", "nerf_mlp_wide.hip")).read()
    prod = src[src.index("DEV void dw2_producer("):src.index("DEV void dw2_consumer(")]
    cons = src[src.index("DEV void dw2_consumer("):src.index("wide_dw2_kernel(")]
    kern = src[src.index("wide_dw2_kernel("):src.index("wide_reduce_kernel(")]
    want = [(str(st), ph) for st in (5, 4, 3, 2, 1) for ph in ("FILLED", "DRAINED")]
    for name, body in (("producer", prod), ("consumer", cons)):
        loop = body[body.index("for (int64_t rd = rd0; rd < rounds; rd += step)"):]
        got = re.findall(r"WD_STAGE_BARRIER\((\d), (FILLED|DRAINED)\)", loop)
        assert got == want, (name, got)
        assert "__syncthreads()" not in body, f"{name}: a barrier outside the named sequence"
    role_a, role_b = kern[kern.index("if (wave < WD_TILES)"):kern.index("} else {")], kern[kern.index("} else {"):]
    assert role_a.count("__syncthreads()") == 1 and role_b.count("__syncthreads()") == 1


def test_slot_fit_key_follows_the_emitter_width_switch():
    """slot scales learned with one emitter width must not be applied to the other - the Python constant that puts
    the width into the fit's key is the kernel file's EQ_WIDE_MIN."""
    import re
    import wisp._C as C
    src = open(os.path.join(ROOT, "kaolin-wisp_amd", "csrc", "hashgrid.hip")).read()
    m = re.search(r"#define EQ_WIDE_MIN \(\(int64_t\)1 << (\d+)\)", src)
    assert m and C.HASHGRID_EMIT_WIDE_MIN == 1 << int(m.group(1))
    import inspect
    assert "HASHGRID_EMIT_WIDE_MIN" in inspect.getsource(C.hashgrid_interpolate_backward)


def test_native_step_config_is_validated_without_a_gpu():
    """wisp_nerf_step_* (csrc/train_step.hip): the struct the binding lays out is the library's, a config with another
    struct_bytes, a table of more than two features or a decoder shape the step does not cover is refused with a message, and
    a well-formed one gets a workspace size that grows with the capacities - all host arithmetic, no device needed."""
    import wisp._C as C
    assert ctypes.sizeof(C.NerfStepConfig) == C.lib.wisp_nerf_step_config_bytes()

    def cfg(**over):
        c = C.NerfStepConfig()
        c.struct_bytes = ctypes.sizeof(C.NerfStepConfig)
        dummy = 0x1000                                     # (never dereferenced by the size query)
        for name in ("octree", "exsum", "table_lookup", "first_idx", "table_grad", "dec_params", "dec_grad", "flat_param", "flat_grad",
                     "flat_exp_avg", "flat_exp_avg_sq"):
            setattr(c, name, dummy)
        keep = ((ctypes.c_int64 * 17)(*range(17)), (ctypes.c_int32 * 16)(*[16] * 16))
        c.first_idx_host, c.resolutions = ctypes.cast(keep[0], ctypes.c_void_p), ctypes.cast(keep[1], ctypes.c_void_p)
        c.level, c.num_samples, c.loss_kind, c.dtype_table, c.num_lods, c.feature_dim, c.bitwidth = 7, 2048, 0, C.BF16, 16, 2, 19
        c.zero_from_col, c.in_dim, c.hidden, c.view_freqs, c.near, c.range = 30, 32, 64, 4, 1.0, 4.0
        c.max_rays, c.max_samples = 4096, 1 << 18
        for k, v in over.items():
            setattr(c, k, v)
        return c, keep

    good, keep = cfg()
    small = C.lib.wisp_nerf_step_workspace_bytes(ctypes.byref(good))
    assert small > (1 << 18) * (8 + 12 + 64 + 64)          # ridx + samples + features + their gradient, at least
    big, keep2 = cfg(max_samples=1 << 20)
    assert C.lib.wisp_nerf_step_workspace_bytes(ctypes.byref(big)) > 2 * small            # (the per-ray buffers do not grow)
    for bad, text in ((dict(struct_bytes=8), "another size"), (dict(feature_dim=4, in_dim=64), "two-feature"), (dict(hidden=128), "decoder shape"),
                      (dict(dtype_table=C.F32), "16-bit"), (dict(range=0.0), "dist_max"), (dict(max_samples=0), "capacities")):
        c, k = cfg(**bad)
        assert C.lib.wisp_nerf_step_workspace_bytes(ctypes.byref(c)) < 0 and text in C.last_error(), (bad, C.last_error())
    assert not C.lib.wisp_nerf_step_create(ctypes.byref(good), None, 0) and "workspace" in C.last_error()


def test_trace_gaps_keeps_only_the_named_regimes_launches(tmp_path):
    """scripts/trace_gaps.py with a regime name: only the launches between that regime's sentinel launches are summarised
    (the 2^18 timeline once held the drop-in loop's steps)."""
    import csv
    import subprocess
    import sys
    import bench
    rows, t = [], [1000]

    def launch(name, dur, grid=256):
        rows.append({"Kernel_Name": name, "Start_Timestamp": t[0], "End_Timestamp": t[0] + dur, "Grid_Size_X": grid, "Workgroup_Size_X": 256})
        t[0] += dur + 500

    def sentinel(tag):
        launch("void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<double>, std::array<char*, 1ul> >", 2000,
               grid=1000 * (tag + 1))
    for tag, fwd_ns in ((0, 30000), (1, 140000)):
        sentinel(tag)
        for _ in range(9):
            launch("void hashgrid_fwd_kernel<__hip_bfloat16, 2, 3>(float const*)", fwd_ns)
            launch("composite_loss_kernel<4>", 5000)
        sentinel(tag)
    d = tmp_path / "trace"
    d.mkdir()
    with open(d / "x_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    for regime, want_us in (("headline", 30.0), ("large_batch_regime", 140.0)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "trace_gaps.py"), str(d), "hashgrid_fwd", "0", regime],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert r.returncode == 0, r.stdout
        line = next(l for l in r.stdout.splitlines() if l.startswith("hashgrid_fwd_kernel"))
        assert abs(float(line.split()[2]) - want_us) < 0.5, (regime, line)
        assert f"regime {regime}: 18 launches" in r.stdout

