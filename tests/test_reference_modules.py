"""CPU: the oracle's decoder / embedder / table layout - and this package's mirrored classes - against the REFERENCE's own
pure-PyTorch modules, executed in place from /root/reference (never copied; the files' `from wisp.core import WispModule`
resolves to this package's class, which is part of what is being checked).  Skipped where the reference tree is not mounted
(the GPU box): everything here is host arithmetic."""
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference/wisp"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")


def _exec_reference(rel):
    """module namespace of a reference source file executed where it lies"""
    path = os.path.join(REF, rel)
    name = "reference_" + rel.replace("/", "_").replace(".py", "")
    mod = types.ModuleType(name)                              # a real module: dataclasses look their module up by name
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile(open(path).read(), path, "exec"), mod.__dict__)
    return mod.__dict__


def test_positional_embedder_oracle_and_mirror_equal_the_reference_module():
    from oracle import nerf as onerf
    from wisp.models.embedders import PositionalEmbedder
    ref = _exec_reference("models/embedders/positional_embedder.py")["PositionalEmbedder"]
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (257, 3)).astype(np.float32))
    for freqs in (4, 10):
        r = ref(freqs, freqs - 1, log_sampling=True, include_input=True, input_dim=3)
        want = r(x)
        assert torch.equal(onerf.positional_embed(x, freqs), want)                       # the oracle's restatement
        mine = PositionalEmbedder(freqs, freqs - 1, log_sampling=True, include_input=True, input_dim=3)
        assert mine.out_dim == r.out_dim == 3 + 6 * freqs
        assert torch.equal(mine(x), want)                                                # the mirrored class
        assert set(mine.state_dict()) == set(r.state_dict())                             # `bands` travels in checkpoints


@pytest.mark.parametrize("bias,layers,hidden", [(True, 1, 64), (False, 1, 128), (True, 2, 32)])
def test_basic_decoder_oracle_and_mirror_equal_the_reference_module(bias, layers, hidden):
    from oracle import nerf as onerf
    from wisp.models.decoders import BasicDecoder
    ref_cls = _exec_reference("models/decoders/basic_decoders.py")["BasicDecoder"]
    torch.manual_seed(3)
    r = ref_cls(32, 16, torch.relu, bias, layer=torch.nn.Linear, num_layers=layers, hidden_dim=hidden, skip=[])
    mine = BasicDecoder(32, 16, torch.relu, bias, layer=torch.nn.Linear, num_layers=layers, hidden_dim=hidden, skip=[])
    orc = onerf.OracleDecoder(32, 16, hidden, layers, bias)
    sd = r.state_dict()
    assert list(mine.state_dict()) == list(sd) == list(orc.state_dict())                 # same names, same order
    mine.load_state_dict(sd)
    orc.load_state_dict(sd)
    x = torch.from_numpy(np.random.default_rng(4).normal(size=(300, 32)).astype(np.float32))
    want = r(x)
    assert torch.equal(mine(x), want) and torch.equal(orc(x), want)
    out, h = r(x, return_h=True)
    out2, h2 = mine(x, return_h=True)
    assert torch.equal(out, out2) and torch.equal(h, h2)


def test_multitable_layout_equals_the_reference_module():
    """MultiTable (grids/utils.py:13-67): per-level sizes min(T, res^dim), begin_idxes, resolutions buffer, the feature
    table's shape and name - and oracle.hashgrid.table_layout."""
    from oracle import hashgrid as ohash
    from wisp.models.grids.utils import MultiTable
    ref_cls = _exec_reference("models/grids/utils.py")["MultiTable"]
    res = [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512]
    for dim, bw in ((3, 19), (3, 12), (2, 14)):
        torch.manual_seed(0)
        r = ref_cls(res, dim, 2, 0.01, 2 ** bw)
        torch.manual_seed(0)
        m = MultiTable(res, dim, 2, 0.01, 2 ** bw)
        sizes, begin = ohash.table_layout(res, 2 ** bw, dim)
        assert [int(v) for v in r.begin_idxes] == [int(v) for v in m.begin_idxes] == [int(v) for v in begin]
        assert list(r.state_dict()) == list(m.state_dict())
        for k, v in r.state_dict().items():
            assert m.state_dict()[k].shape == v.shape and m.state_dict()[k].dtype == v.dtype, k
        assert torch.equal(r.resolutions.reshape(-1).long(), m.resolutions.reshape(-1).long())
        assert r.feats.shape == m.feats.shape == (int(begin[-1]), 2)
        assert torch.equal(r.feats, m.feats)                                             # same draw order: randn(total, F) * std


def test_radiance_field_rgba_arithmetic_equals_reference_decoder_chain():
    """NeuralRadianceField.rgba (nerf.py:219-264) restated with the reference's OWN embedder and decoders on the oracle's
    grid features: density = relu(y[..., 0:1]), colour = sigmoid(decoder_color(cat(y, embed(dir))[..., 1:])) - what
    oracle.nerf computes after its hash-grid lookup and what the fused HIP decoder is tested against."""
    from oracle import hashgrid as ohash, nerf as onerf
    emb_cls = _exec_reference("models/embedders/positional_embedder.py")["PositionalEmbedder"]
    dec_cls = _exec_reference("models/decoders/basic_decoders.py")["BasicDecoder"]
    torch.manual_seed(9)
    emb = emb_cls(4, 3, log_sampling=True, include_input=True, input_dim=3)
    dd = dec_cls(32, 16, torch.relu, True, layer=torch.nn.Linear, num_layers=1, hidden_dim=64, skip=[])
    dc = dec_cls(15 + emb.out_dim, 3, torch.relu, True, layer=torch.nn.Linear, num_layers=2, hidden_dim=64, skip=[])   # nerf.py:165-173
    res = [16, 32, 64, 128, 256, 300, 350, 400, 420, 440, 460, 470, 480, 490, 500, 512]
    onef = onerf.OracleNeRF(res, 2, 10, 'cat', 0.1, 64, 1, True, 4)
    assert onef.view_embed_dim == emb.out_dim
    sd = {("decoder_density." + k): v for k, v in dd.state_dict().items()}
    sd.update({("decoder_color." + k): v for k, v in dc.state_dict().items()})
    info = onef.load_state_dict(sd, strict=False)
    assert not info.unexpected_keys and all(not k.startswith("decoder") for k in info.missing_keys)
    rng = np.random.default_rng(10)
    coords = torch.from_numpy(rng.uniform(-1, 1, (200, 3)).astype(np.float32))
    dirs = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(200, 3)).astype(np.float32)), dim=1)
    with torch.no_grad():
        got = onef.rgba(coords, dirs)
        feats = ohash.grid_interpolate(coords, len(res) - 1, 'cat', 2, res, 10, onef.grid.codebook.feats, onef.begin_idxes)
        y = dd(feats.float())
        fdir = torch.cat([y, emb(dirs)], dim=-1)
        want_rgb = torch.sigmoid(dc(fdir[..., 1:]))
        want_density = torch.relu(y[..., 0:1])
    assert torch.equal(got["density"], want_density) and torch.equal(got["rgb"], want_rgb)


def _same_rays(a, b):
    assert type(a).__name__ == type(b).__name__ == "Rays"
    assert torch.equal(a.origins, b.origins) and torch.equal(a.dirs, b.dirs)
    for k in ("dist_min", "dist_max"):
        x, y = getattr(a, k), getattr(b, k)
        assert (torch.equal(x, y) if torch.is_tensor(x) else x == y), k


def test_rays_container_behaves_like_the_reference_class():
    """wisp.core.Rays against the reference's own dataclass (wisp/core/rays.py) over the whole method surface."""
    from wisp.core import Rays as Mine
    Ref = _exec_reference("core/rays.py")["Rays"]
    rng = np.random.default_rng(20)
    o = torch.from_numpy(rng.normal(size=(6, 4, 3)).astype(np.float32))
    d = torch.from_numpy(rng.normal(size=(6, 4, 3)).astype(np.float32))
    for kw in (dict(dist_min=0.5, dist_max=7.0), dict(dist_min=torch.rand(6, 4, 1), dist_max=torch.rand(6, 4, 1) + 2)):
        r, m = Ref(o, d, **kw), Mine(o, d, **kw)
        assert len(r) == len(m) and tuple(r.shape) == tuple(m.shape) and r.ndim == m.ndim
        _same_rays(r[2:5], m[2:5])
        _same_rays(r.reshape(-1, 3), m.reshape(-1, 3))
        _same_rays(r.contiguous(), m.contiguous())
        _same_rays(r.to(torch.float64), m.to(torch.float64))
        if not torch.is_tensor(kw["dist_min"]):               # the reference's cat / stack take min() / max() of the bounds: scalars only
            _same_rays(Ref.cat([r, r], dim=0), Mine.cat([m, m], dim=0))
            _same_rays(Ref.stack([r, r], dim=0), Mine.stack([m, m], dim=0))
        for x, y in zip(r.split(4), m.split(4)):
            _same_rays(x, y)
        _same_rays(r[:, 0:1].squeeze(1), m[:, 0:1].squeeze(1))


def test_spc_sampling_helpers_equal_the_reference_functions():
    """wisp/ops/spc/sampling.py:35-71 (kept for API compatibility; the tracer's hot path is fused in csrc/raymarch.hip)."""
    from wisp.ops.spc import sampling as mine
    ref = _exec_reference("ops/spc/sampling.py")
    rng = np.random.default_rng(21)
    entry = torch.from_numpy(rng.uniform(0, 3, (50, 1)).astype(np.float32))
    iv = torch.cat([entry, entry + torch.from_numpy(rng.uniform(0.01, 0.3, (50, 1)).astype(np.float32))], 1)
    torch.manual_seed(5)
    want = ref["sample_from_depth_intervals"](iv, 16)
    torch.manual_seed(5)
    got = mine.sample_from_depth_intervals(iv, 16)
    assert torch.equal(got, want)                                                        # same draws, same arithmetic
    flags = torch.from_numpy(rng.integers(0, 2, 50).astype(bool))
    assert torch.equal(mine.expand_pack_boundary(flags, 16), ref["expand_pack_boundary"](flags, 16))


def _kaolin_stub():
    """the one Kaolin leaf HashGrid.__init__ calls: unbatched_get_level_points = the slice of the point hierarchy of a level
    (pyramid row 0 = counts, row 1 = offsets; kaolin/ops/spc/spc.py)"""
    k = types.ModuleType("kaolin"); k.ops = types.ModuleType("kaolin.ops"); k.ops.spc = types.ModuleType("kaolin.ops.spc")
    k.ops.spc.unbatched_get_level_points = lambda points, pyramid, level: points[int(pyramid[1, level]):int(pyramid[1, level]) + int(pyramid[0, level])]
    k._C = types.ModuleType("kaolin._C")                      # `from kaolin import _C` (wisp/ops/grid.py:10): imported, unused here
    return {"kaolin": k, "kaolin.ops": k.ops, "kaolin.ops.spc": k.ops.spc, "kaolin._C": k._C}


def test_hash_grid_constructors_equal_the_reference_class():
    """The reference's OWN HashGrid class (models/grids/hash_grid.py:28-200), executed in place over this package's BLAS,
    BLASGrid base and ops - and over the reference's own MultiTable - against wisp.models.grids.HashGrid: same level
    resolutions from from_geometric / from_octree, same bookkeeping attributes, same state dict, same dense cell list."""
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import HashGrid as Mine
    import wisp.models.grids.utils as my_utils
    ref_utils = _exec_reference("models/grids/utils.py")
    stubs = _kaolin_stub()
    saved = {k: sys.modules.get(k) for k in stubs}
    saved_table = my_utils.MultiTable
    sys.modules.update(stubs)
    my_utils.MultiTable = ref_utils["MultiTable"]              # `from wisp.models.grids.utils import MultiTable` inside the reference file
    try:
        Ref = _exec_reference("models/grids/hash_grid.py")["HashGrid"]
    finally:
        my_utils.MultiTable = saved_table
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    blas = OctreeAS.make_dense(3)
    cases = [("from_geometric", dict(feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=0.01, codebook_bitwidth=19,
                                     min_grid_res=16, max_grid_res=512)),
             ("from_geometric", dict(feature_dim=4, num_lods=5, multiscale_type='sum', feature_std=0.1, codebook_bitwidth=12,
                                     min_grid_res=8, max_grid_res=300)),
             ("from_octree", dict(feature_dim=2, base_lod=3, num_lods=4, multiscale_type='cat', feature_std=0.01, codebook_bitwidth=14))]
    for ctor, kw in cases:
        torch.manual_seed(0)
        r = getattr(Ref, ctor)(blas, **kw)
        torch.manual_seed(0)
        m = getattr(Mine, ctor)(blas, **kw)
        assert [int(x) for x in r.resolutions] == [int(x) for x in m.resolutions], ctor
        for attr in ("num_lods", "max_lod", "active_lods", "codebook_size", "codebook_bitwidth", "feature_dim", "multiscale_type",
                     "coord_dim", "num_cells"):
            assert getattr(r, attr) == getattr(m, attr), (ctor, attr)
        assert torch.equal(r.dense_points.long(), m.dense_points.long()) and r.occupancy.shape == m.occupancy.shape
        rs, ms = r.state_dict(), m.state_dict()
        assert list(rs) == list(ms), (list(rs), list(ms))
        for k in rs:
            assert rs[k].shape == ms[k].shape and rs[k].dtype == ms[k].dtype, k
        assert torch.equal(rs["codebook.feats"], ms["codebook.feats"])                   # same initial draw
        assert r.blas is m.blas is blas


def test_accelstruct_result_dataclasses_equal_the_reference_definitions():
    """AS*Results (accelstructs/base_as.py): field names, order and defaults are the contract between BLAS and tracers."""
    import dataclasses
    import wisp.accelstructs as mine
    ref = _exec_reference("accelstructs/base_as.py")
    for name in ("ASQueryResults", "ASRaytraceResults", "ASRaymarchResults"):
        rf, mf = dataclasses.fields(ref[name]), dataclasses.fields(getattr(mine, name))
        assert [f.name for f in rf] == [f.name for f in mf], name
        for a, b in zip(rf, mf):
            assert (a.default is dataclasses.MISSING) == (b.default is dataclasses.MISSING), (name, a.name)
            if a.default is not dataclasses.MISSING:
                assert a.default == b.default, (name, a.name)
    for meth in ("query", "raytrace", "raymarch", "occupancy", "capacity_at", "name"):
        if hasattr(ref["BaseAS"], meth):
            assert hasattr(mine.BaseAS, meth), meth


def test_sphere_samplers_equal_the_reference_functions():
    """wisp/ops/geometric.py:25-62 - the prune's view directions come from sample_unif_sphere (numpy's global generator)."""
    import wisp.ops.geometric as mine
    ref = _exec_reference("ops/geometric.py")             # its `import wisp._C` is this package's module
    np.random.seed(4)
    want = ref["sample_unif_sphere"](1000)
    np.random.seed(4)
    got = mine.sample_unif_sphere(1000)
    assert np.array_equal(got, want) and got.shape == (1000, 3)
    assert np.array_equal(mine.sample_fib_sphere(777), ref["sample_fib_sphere"](777))


def test_radiance_field_construction_equals_the_reference_class():
    """The reference's OWN NeuralRadianceField (models/nefs/nerf.py:30-217), executed in place over this package's grids,
    decoders, embedders and base class, next to wisp.models.nefs.NeuralRadianceField built with the same arguments: same
    decoders (names, order, shapes, initial values incl. the density bias of nerf.py:162-163), same registered channels,
    same feature / embedding widths - for the nerf_hash.yaml arguments and a positional-position, bias-free variant."""
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField as Mine
    Ref = _exec_reference("models/nefs/nerf.py")["NeuralRadianceField"]
    blas = OctreeAS.make_dense(3)
    for kw in (dict(pos_embedder='none', view_embedder='positional', view_multires=4, activation_type='relu', layer_type='linear',
                    hidden_dim=64, num_layers=1, bias=True, prune_density_decay=0.95, prune_min_density=2.956),
               dict(pos_embedder='positional', pos_multires=6, view_embedder='positional', view_multires=2, activation_type='relu',
                    layer_type='linear', hidden_dim=128, num_layers=2, bias=False)):
        torch.manual_seed(1)
        grid_r = HashGrid.from_geometric(blas, feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=0.01,
                                         codebook_bitwidth=12, min_grid_res=16, max_grid_res=512)
        r = Ref(grid_r, **kw)
        torch.manual_seed(1)
        grid_m = HashGrid.from_geometric(blas, feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=0.01,
                                         codebook_bitwidth=12, min_grid_res=16, max_grid_res=512)
        m = Mine(grid_m, **kw)
        rs, ms = r.state_dict(), m.state_dict()
        assert list(rs) == list(ms)
        for k in rs:
            assert torch.equal(rs[k], ms[k]), k                                          # same construction order => same draws
        assert r.effective_feature_dim() == m.effective_feature_dim()
        assert r.view_embed_dim == m.view_embed_dim and r.pos_embed_dim == m.pos_embed_dim
        assert r.get_supported_channels() == m.get_supported_channels() == {"rgb", "density"}
        for attr in ("prune_density_decay", "prune_min_density", "hidden_dim", "num_layers", "bias", "position_input"):
            if hasattr(r, attr):
                assert getattr(r, attr) == getattr(m, attr), attr


def test_base_tracer_argument_plumbing_equals_the_reference_class():
    """BaseTracer.forward (tracers/base_tracer.py:99-162): channel negotiation and how trace()'s optional arguments are filled
    from keyword arguments or from same-named attributes - the reference's class and this package's, subclassed identically."""
    from wisp.tracers import BaseTracer as Mine
    Ref = _exec_reference("tracers/base_tracer.py")["BaseTracer"]

    def make(base):
        class T(base):
            def __init__(self):
                super().__init__(bg_color=(0.1, 0.2, 0.3))
                self.num_steps, self.raymarch_type, self.step_size = 96, 'voxel', None

            def get_supported_channels(self):
                return {"rgb", "depth"}

            def get_required_nef_channels(self):
                return {"rgb", "density"}

            def trace(self, nef, rays, channels, extra_channels, lod_idx=None, raymarch_type='ray', num_steps=64, step_size=1.0, bg_color=None):
                return dict(channels=channels, extra=extra_channels, lod_idx=lod_idx, raymarch_type=raymarch_type,
                            num_steps=num_steps, step_size=step_size, bg_color=bg_color)
        return T()

    class Nef:
        def __init__(self, chans):
            self.chans = chans

        def get_supported_channels(self):
            return set(self.chans)

    r, m = make(Ref), make(Mine)
    nef = Nef({"rgb", "density", "normal"})
    for kw in (dict(), dict(channels="rgb"), dict(channels=["rgb", "normal"]), dict(channels={"depth"}, num_steps=7, lod_idx=2),
               dict(channels=None, raymarch_type='uniform', step_size=0.5)):
        assert r(nef, "rays", **kw) == m(nef, "rays", **kw), kw
    for bad_nef, kw in ((Nef({"rgb"}), dict()), (nef, dict(channels=["rgb", "semantics"]))):
        with pytest.raises(Exception) as e1:
            r(bad_nef, "rays", **kw)
        with pytest.raises(Exception) as e2:
            m(bad_nef, "rays", **kw)
        assert type(e1.value) is type(e2.value)


def _with_kaolin_stub(fn):
    stubs = _kaolin_stub()
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        return fn()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _same_module_state(r, m, what):
    rs, ms = r.state_dict(), m.state_dict()
    assert list(rs) == list(ms), (what, list(rs), list(ms))
    for k in rs:
        assert rs[k].shape == ms[k].shape and rs[k].dtype == ms[k].dtype, (what, k)
        assert torch.equal(rs[k], ms[k]), (what, k)


@pytest.mark.parametrize("kind", ["octree", "codebook"])
def test_octree_grid_constructors_equal_the_reference_classes(kind):
    """The reference's OWN OctreeGrid / CodebookOctreeGrid (models/grids/octree_grid.py:25-120, codebook_grid.py:25-100),
    executed in place over this package's BLAS and `wisp.ops.spc.make_trilinear_spc`, against the mirrored classes: feature
    pyramid (corner counts per level + 1), parameter names, shapes and initial draws, active levels."""
    from wisp.accelstructs import OctreeAS
    import wisp.models.grids as mine
    rng = np.random.default_rng(50)
    pts = torch.from_numpy(rng.integers(0, 16, size=(300, 3)).astype(np.int16))
    blas = OctreeAS.from_quantized_points(pts, 4)
    if kind == "octree":
        Ref = _with_kaolin_stub(lambda: _exec_reference("models/grids/octree_grid.py")["OctreeGrid"])
        Mine, kw = mine.OctreeGrid, dict(feature_dim=5, num_lods=3, interpolation_type='linear', multiscale_type='sum', feature_std=0.1)
    else:
        Ref = _with_kaolin_stub(lambda: _exec_reference("models/grids/codebook_grid.py")["CodebookOctreeGrid"])
        Mine, kw = mine.CodebookOctreeGrid, dict(feature_dim=5, num_lods=3, interpolation_type='linear', multiscale_type='sum',
                                                 feature_std=0.1, codebook_bitwidth=4)
    torch.manual_seed(2)
    r = Ref(blas, **kw)
    torch.manual_seed(2)
    m = Mine(blas, **kw)
    _same_module_state(r, m, kind)
    for attr in ("feature_dim", "max_lod", "num_lods", "base_lod", "active_lods", "interpolation_type", "multiscale_type"):
        assert getattr(r, attr) == getattr(m, attr), attr
    assert torch.equal(r.trinkets.long(), m.trinkets.long()) and torch.equal(r.pyramid_dual.long(), m.pyramid_dual.long())


def test_neural_sdf_and_pipeline_construction_equal_the_reference_classes():
    """NeuralSDF (models/nefs/neural_sdf.py:22-120) and Pipeline (models/pipeline.py) from the reference's own sources over
    this package's grid: the nglod_octree.yaml decoder (19 -> 128 -> 1) and the channel registration."""
    from wisp.accelstructs import OctreeAS
    from wisp.models import Pipeline as MinePipe
    from wisp.models.grids import OctreeGrid
    from wisp.models.nefs import NeuralSDF as Mine
    Ref = _exec_reference("models/nefs/neural_sdf.py")["NeuralSDF"]
    RefPipe = _exec_reference("models/pipeline.py")["Pipeline"]
    rng = np.random.default_rng(51)
    blas = OctreeAS.from_quantized_points(torch.from_numpy(rng.integers(0, 16, size=(200, 3)).astype(np.int16)), 4)
    for kw in (dict(pos_embedder='none', position_input=True, activation_type='relu', layer_type='none', hidden_dim=128, num_layers=1),
               # ('positional' cannot be compared: the reference's own NeuralSDF.init_embedder passes a keyword its
               #  get_positional_embedder does not take, neural_sdf.py:97 - it raises TypeError in the reference itself)
               dict(pos_embedder='identity', position_input=False, hidden_dim=64, num_layers=2)):
        torch.manual_seed(3)
        r = Ref(OctreeGrid(blas, feature_dim=16, num_lods=3, multiscale_type='sum', feature_std=0.01), **kw)
        torch.manual_seed(3)
        m = Mine(OctreeGrid(blas, feature_dim=16, num_lods=3, multiscale_type='sum', feature_std=0.01), **kw)
        _same_module_state(r, m, "NeuralSDF")
        assert r.get_supported_channels() == m.get_supported_channels() == {"sdf"}
        assert r.decoder_input_dim() == m.decoder_input_dim() if hasattr(r, "decoder_input_dim") else True

    class Tracer(torch.nn.Module):
        def forward(self, nef, **kw):
            return ("traced", nef, kw)
    pr, pm = RefPipe(m, Tracer()), MinePipe(m, Tracer())
    assert pr(rays="R", channels=["sdf"])[2] == pm(rays="R", channels=["sdf"])[2]
    assert [n for n, _ in pr.named_children()] == [n for n, _ in pm.named_children()]


def _reference_method(rel, cls_name, meth_name, glb):
    """compile ONE method of a reference class from the file where it lies (the module itself imports the whole application)"""
    import ast
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read(), path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name)
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == meth_name)
    fn.decorator_list = []                                    # profiler ranges (@torch.cuda.nvtx.range): not part of the semantics
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = dict(glb)
    exec(compile(mod, path, "exec"), ns)
    return ns[meth_name]


def test_optimizer_parameter_groups_equal_the_reference_init_optimizer():
    """BaseTrainer.init_optimizer (trainers/base_trainer.py:205-236), the method itself compiled from the reference file and
    run on a stand-in trainer: which parameter lands in which group with which learning rate / weight decay - against the
    flat-buffer groups MultiviewTrainStep hands its fused optimizer."""
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.trainers import FlatParams
    init_optimizer = _reference_method("trainers/base_trainer.py", "BaseTrainer", "init_optimizer",
                                       dict(torch=torch, instantiate=lambda cfg, params: torch.optim.AdamW(
                                           params, lr=cfg.lr, eps=cfg.eps, weight_decay=cfg.weight_decay, betas=cfg.betas)))
    torch.manual_seed(0)
    grid = HashGrid.from_geometric(OctreeAS.make_dense(3), feature_dim=2, num_lods=8, multiscale_type='cat', feature_std=0.01,
                                   codebook_bitwidth=10, min_grid_res=8, max_grid_res=64)
    nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1, bias=True)
    cfg = types.SimpleNamespace(optimizer=types.SimpleNamespace(lr=1e-3, eps=1e-16, weight_decay=1e-6, betas=(0.9, 0.999)),
                                grid_lr_weight=100.0, max_epochs=10, scheduler=False, scheduler_milestones=[0.5], scheduler_gamma=0.333)
    me = types.SimpleNamespace(pipeline=types.SimpleNamespace(nef=nef), cfg=cfg, train_dataset=list(range(7)))
    init_optimizer(me)
    groups = me.optimizer.param_groups
    assert len(groups) == 3
    by_ptr = {}
    for g in groups:
        for p in g["params"]:
            by_ptr[p.data_ptr()] = (g["lr"], g["weight_decay"], g["eps"])
    # this package: one flat buffer, ranges per group; MultiviewTrainStep.optimizer_step gives range g (lr_g, weight_decay, eps)
    names = {p.data_ptr(): n for n, p in nef.named_parameters()}
    want = {n: by_ptr[ptr] for ptr, n in names.items()}
    flat = FlatParams(nef)
    lr_of = {"decoder": cfg.optimizer.lr, "grid": cfg.optimizer.lr * cfg.grid_lr_weight, "rest": cfg.optimizer.lr}
    base = flat.data.data_ptr()
    seen = 0
    for n, p in nef.named_parameters():
        if not p.requires_grad:                            # the embedder's frozen `bands`: in the reference's 'rest' group, never stepped
            assert want[n][0] == cfg.optimizer.lr
            continue
        off = (p.data_ptr() - base) // 4
        grp = next(g for g, (a, b) in flat.ranges.items() if a <= off < b)
        assert (lr_of[grp], cfg.optimizer.weight_decay, cfg.optimizer.eps) == want[n], (n, grp)
        seen += 1
    assert seen == sum(1 for _, p in nef.named_parameters() if p.requires_grad) >= 9
    assert abs(want["grid.codebook.feats"][0] - 0.1) < 1e-12


class _TorchWithoutNvtx:
    """`torch` for reference method bodies on a CPU box: everything forwarded, torch.cuda.nvtx.range a no-op context"""
    def __init__(self):
        import contextlib
        self.cuda = types.SimpleNamespace(nvtx=types.SimpleNamespace(range=lambda *_a, **_k: contextlib.nullcontext()),
                                          amp=torch.cuda.amp)

    def __getattr__(self, name):
        return getattr(torch, name)


@pytest.mark.parametrize("loss_type", ["huber", "l2", "l1"])
def test_training_step_semantics_equal_the_reference_methods(loss_type):
    """MultiviewTrainer.pre_step / step / calc_adaptive_rays (trainers/multiview_trainer.py:86-180): the three method bodies
    compiled from the reference file and driven for six iterations (prune_every = 2: two prunes) over a CPU stand-in
    pipeline with torch.optim.AdamW - next to MultiviewTrainStep.step over an identical pipeline (its fused optimizer replaced
    by the CPU restatement of the kernel arithmetic).  Same prune timing, same loss, same parameters afterwards, same adaptive
    ray count."""
    import math
    import random
    import test_distributed_gloo as stub
    import wisp._C as C
    from wisp.core import Rays
    from wisp.datasets.transforms import SampleRays
    from wisp.trainers import MultiviewTrainStep
    glb = dict(torch=_TorchWithoutNvtx(), math=math, random=random, SampleRays=SampleRays)
    ref_pre = _reference_method("trainers/multiview_trainer.py", "MultiviewTrainer", "pre_step", glb)
    ref_step = _reference_method("trainers/multiview_trainer.py", "MultiviewTrainer", "step", glb)
    ref_rays = _reference_method("trainers/multiview_trainer.py", "MultiviewTrainer", "calc_adaptive_rays", glb)
    g = torch.Generator().manual_seed(3)
    O, D, T = torch.rand(96, 3, generator=g) * 2 - 1, torch.randn(96, 3, generator=g), torch.rand(96, 3, generator=g)
    lr, wd, eps, glw = 1e-2, 1e-6, 1e-16, 10.0

    # ---- the reference's methods on a stand-in trainer object
    pipe_r = stub._StubPipeline()
    pipe_r.tracer.raymarch_type, pipe_r.tracer.num_steps = 'ray', 64
    prunes_r = []
    pipe_r.nef.prune = lambda: prunes_r.append(me.total_iterations)
    named = dict(pipe_r.nef.named_parameters())
    groups = [{"params": [p for n, p in named.items() if 'decoder' in n], "lr": lr},
              {"params": [p for n, p in named.items() if 'decoder' not in n and 'grid' in n], "lr": lr * glw},
              {"params": [p for n, p in named.items() if 'decoder' not in n and 'grid' not in n], "lr": lr}]
    metrics = types.SimpleNamespace(total_loss=0.0, rgb_loss=0.0, num_samples=0)
    me = types.SimpleNamespace(
        pipeline=pipe_r, device='cpu', total_iterations=0, optimizer=torch.optim.AdamW(groups, lr=lr, eps=eps, weight_decay=wd),
        cfg=types.SimpleNamespace(prune_every=2, random_lod=False, rgb_loss_type=loss_type, rgb_loss_denom='rays', opacity_loss=0.0,
                                  enable_amp=False, scheduler=False, target_sample_size=2 ** 12),
        tracker=types.SimpleNamespace(metrics=metrics), train_dataset=types.SimpleNamespace(transform=SampleRays(96)))
    me.calc_adaptive_rays = lambda rays, warmup=False: ref_rays(me, rays, warmup)

    class _Base:                                              # super().pre_step() of BaseTrainer: nothing this path reads
        def pre_step(self):
            pass
    glb["super"] = lambda: _Base()
    ref_pre = _reference_method("trainers/multiview_trainer.py", "MultiviewTrainer", "pre_step", glb)
    ref_losses = []
    for it in range(6):
        ref_pre(me)
        before = metrics.rgb_loss
        ref_step(me, {"rays": Rays(O, D)[None] if False else Rays(O[None], D[None]), "rgb": T[None]})
        ref_losses.append(metrics.rgb_loss - before)
        me.total_iterations += 1                              # BaseTrainer.iterate: post_step bookkeeping

    # ---- this package's step
    C.adamw_step_groups = stub._torch_adamw_groups
    pipe_m = stub._StubPipeline()
    tr = MultiviewTrainStep(pipe_m, lr=lr, eps=eps, weight_decay=wd, grid_lr_weight=glw, rgb_loss_type=loss_type, prune_every=2,
                            target_sample_size=2 ** 12, seed=5)
    my_losses = []
    for it in range(6):
        loss, _ = tr.step(Rays(O, D), T)
        my_losses.append(float(loss))
    assert prunes_r == [2, 4] and len(pipe_m.nef.prune_log) == 2                          # prune before iterations 2 and 4 on both sides
    np.testing.assert_allclose(my_losses, ref_losses, rtol=2e-6, atol=1e-8)
    for (n1, p1), (n2, p2) in zip(pipe_r.nef.named_parameters(), pipe_m.nef.named_parameters()):
        assert n1 == n2
        np.testing.assert_allclose(p2.detach().numpy(), p1.detach().numpy(), rtol=1e-5, atol=2e-7, err_msg=n1)
    assert tr.num_rays == me.train_dataset.transform.num_samples                          # same adaptive ray count


@pytest.mark.parametrize("amp", [False, True])
def test_dropin_trainer_class_equals_the_reference_methods(amp):
    """wisp.trainers.MultiviewTrainer (the unchanged-application regime: torch.optim groups from init_optimizer, GradScaler,
    per-step metric read-backs, MultiStepLR) driven through its own BaseTrainer.iterate() - next to a trainer assembled from
    the reference's OWN method bodies, compiled from the files where they lie: BaseTrainer.iterate / begin_epoch / end_epoch /
    init_optimizer (base_trainer.py:205-342) and MultiviewTrainer.pre_step / step / calc_adaptive_rays
    (multiview_trainer.py:85-180).  Same iteration numbering across epoch boundaries (the first step of a new epoch runs with
    iteration 0), warm-up call, prune timing, losses, learning-rate schedule, parameters and adaptive ray count.  (On this
    CPU box the GradScaler of either side is disabled, as torch does without a GPU; the enable_amp branch is still the one
    executed.  The fp16 + scaler arithmetic itself runs under -m gpu.)"""
    import math
    import random
    import time
    import test_distributed_gloo as stub
    from wisp.core import Rays
    from wisp.datasets.transforms import SampleRays
    from wisp.trainers import MultiviewTrainer, ConfigMultiviewTrainer, ConfigAdamW
    g = torch.Generator().manual_seed(3)
    V, P, STEPS = 5, 96, 13
    O, D, T = torch.rand(V, P, 3, generator=g) * 2 - 1, torch.randn(V, P, 3, generator=g), torch.rand(V, P, 3, generator=g)

    class _Views:                                              # whole views in a fixed order: both sides see identical batches
        def __init__(self):
            self.transform = SampleRays(P)
        def __len__(self):
            return V
    class _Loader:
        def __len__(self):
            return V
        def __iter__(self):
            return iter([{"rays": Rays(O[i][None], D[i][None]), "rgb": T[i][None]} for i in range(V)])

    def make_cfg():
        return ConfigMultiviewTrainer(optimizer=ConfigAdamW(lr=1e-2, eps=1e-16, weight_decay=1e-6), grid_lr_weight=10.0,
                                      max_epochs=3, enable_amp=amp, scheduler=True, scheduler_milestones=(0.3, 0.6),
                                      scheduler_gamma=0.5, prune_every=2, rgb_loss_type='huber', target_sample_size=2 ** 12)

    def make_pipe():
        pipe = stub._StubPipeline()
        pipe.tracer.raymarch_type, pipe.tracer.num_steps, pipe.tracer.prev_num_samples = 'ray', 64, None
        pipe.nef.grid.raymarch = lambda rays, **kw: types.SimpleNamespace(samples=torch.zeros(64 * 7, 3))
        pipe.nef.grid.active_lods = [0]
        return pipe

    # ---- this package's class through its own life cycle
    pipe_m, prunes_m = make_pipe(), []
    tr = MultiviewTrainer(make_cfg(), pipe_m, _Views(), device='cpu')
    tr.train_data_loader = _Loader()
    pipe_m.nef.prune = lambda: prunes_m.append(tr.total_iterations)
    tr.is_optimization_running = True
    losses_m, lrs_m, its_m = [], [], []
    for _ in range(STEPS):
        before = tr.tracker.metrics.rgb_loss
        tr.iterate()
        its_m.append((tr.epoch, tr.iteration))
        losses_m.append(tr.tracker.metrics.rgb_loss - before)
        lrs_m.append([g_["lr"] for g_ in tr.optimizer.param_groups])

    # ---- a trainer made of the reference's method bodies
    glb = dict(torch=_TorchWithoutNvtx(), math=math, random=random, SampleRays=SampleRays, time=time,
               instantiate=lambda cfg, params: torch.optim.AdamW(params, lr=cfg.lr, eps=cfg.eps, weight_decay=cfg.weight_decay,
                                                                 betas=cfg.betas))
    class _Base:
        def pre_step(self):
            pass
    glb["super"] = lambda: _Base()
    body = {}
    for m in ("iterate", "begin_epoch", "end_epoch", "is_first_iteration", "is_any_iterations_remaining", "reset_data_iterator",
              "next_batch", "init_optimizer"):
        body[m] = _reference_method("trainers/base_trainer.py", "BaseTrainer", m, glb)
    for m in ("pre_step", "step", "calc_adaptive_rays"):
        body[m] = _reference_method("trainers/multiview_trainer.py", "MultiviewTrainer", m, glb)
    noop = lambda self, *a, **k: None
    RefTrainer = type("RefTrainer", (), dict(
        body, pre_training=noop, post_training=noop, post_epoch=noop, post_step=noop, validate=noop,
        pre_epoch=lambda self: self.tracker.metrics.__dict__.update(total_loss=0.0, rgb_loss=0.0, num_samples=0),
        total_iterations=property(lambda self: (self.epoch - 1) * self.iterations_per_epoch + self.iteration),   # base_trainer.py:562-567
        max_iterations=property(lambda self: self.max_epochs * self.iterations_per_epoch)))                       # :583-586
    pipe_r, prunes_r = make_pipe(), []
    me = RefTrainer()
    me.pipeline, me.device, me.cfg, me.enable_amp = pipe_r, 'cpu', make_cfg(), amp
    me.tracker = types.SimpleNamespace(metrics=types.SimpleNamespace(total_loss=0.0, rgb_loss=0.0, num_samples=0),
                                       log_metric=lambda *a, **k: None)
    me.scene_state = types.SimpleNamespace(optimization=types.SimpleNamespace(elapsed_time=0.0, iterations_per_epoch=V))
    me.train_dataset = _Views()
    me.train_data_loader, me.train_data_loader_iter = _Loader(), None
    me.epoch, me.iteration, me.iterations_per_epoch, me.max_epochs = 1, 0, V, me.cfg.max_epochs
    me.is_optimization_running = True
    me.init_optimizer()
    pipe_r.nef.prune = lambda: prunes_r.append(me.total_iterations)
    losses_r, lrs_r, its_r = [], [], []
    for _ in range(STEPS):
        before = me.tracker.metrics.rgb_loss
        me.iterate()
        its_r.append((me.epoch, me.iteration))
        losses_r.append(me.tracker.metrics.rgb_loss - before)
        lrs_r.append([g_["lr"] for g_ in me.optimizer.param_groups])

    assert its_m == its_r and (2, 0) in its_r                        # iteration numbering incl. the epoch boundary quirk
    assert prunes_m == prunes_r and len(prunes_r) >= 4
    assert losses_m[0] == 0.0 and losses_r[0] == 0.0                 # the warm-up call optimises nothing
    np.testing.assert_allclose(losses_m, losses_r, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(lrs_m, lrs_r, rtol=1e-12)
    assert lrs_m[0] != lrs_m[-1]                                     # the schedule did fire
    for (n1, p1), (n2, p2) in zip(pipe_r.nef.named_parameters(), pipe_m.nef.named_parameters()):
        assert n1 == n2
        np.testing.assert_allclose(p2.detach().numpy(), p1.detach().numpy(), rtol=1e-6, atol=1e-8, err_msg=n1)
    assert tr.train_dataset.transform.num_samples == me.train_dataset.transform.num_samples
    assert [g_["weight_decay"] for g_ in tr.optimizer.param_groups] == [g_["weight_decay"] for g_ in me.optimizer.param_groups]


def _torch_optim_groups(kind, param, grad, s1, s2, groups, h0, h1, eps, step, grad_scale=1.0, zero_grad=False):
    """CPU stand-in for the fused optimizer launch (test infrastructure): csrc/misc.hip optim_groups_kernel, kind 'adam'."""
    assert kind == 'adam'
    bc1, bc2 = 1 - h0 ** step, (1 - h1 ** step) ** 0.5
    for a, n, lr, wd, _shadow in groups:
        p, g, m, v = param[a:a + n], grad[a:a + n] * grad_scale, s1[a:a + n], s2[a:a + n]
        g = g + wd * p
        m.mul_(h0).add_(g, alpha=1 - h0)
        v.mul_(h1).addcmul_(g, g, value=1 - h1)
        p.sub_((lr / bc1) * m / (v.sqrt() / bc2 + eps))
    if zero_grad:
        grad.zero_()


def test_sdf_training_step_semantics_equal_the_reference_method():
    """SDFTrainer.step (trainers/sdf_trainer.py:65-124), the method body compiled from the reference file, over a CPU
    stand-in field with torch.optim.Adam - next to SDFTrainStep.step (fused optimizer replaced by its CPU restatement):
    loss = sum over the loss LODs of the squared error, divided by the batch size; same parameters after five steps."""
    import wisp._C as C
    from wisp.trainers import SDFTrainStep
    ref_step = _reference_method("trainers/sdf_trainer.py", "SDFTrainer", "step", dict(torch=_TorchWithoutNvtx()))

    class Field(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(8)
            self.grid = torch.nn.Module()
            self.grid.num_lods = 3
            self.grid.feats = torch.nn.Parameter(torch.randn(3, 32, 4) * 0.1)            # 'grid' in the name -> grid group
            self.decoder = torch.nn.Linear(4 + 3, 1)

        def forward(self, coords=None, lod_idx=None, channels=None):
            cell = ((coords[:, 0] * 0.5 + 0.5) * 31).long().clamp(0, 31)
            f = self.grid.feats[: lod_idx + 1, cell].sum(0)
            out = self.decoder(torch.cat([f, coords], -1))
            return [out] if isinstance(channels, (list, tuple)) else out

    g = torch.Generator().manual_seed(2)
    X, Y = torch.rand(5, 64, 3, generator=g) * 2 - 1, torch.randn(5, 64, 1, generator=g) * 0.1
    for only_last in (True, False):
        fr, fm = Field(), Field()
        lods = [2] if only_last else [0, 1, 2]
        named = dict(fr.named_parameters())
        opt = torch.optim.Adam([{"params": [p for n, p in named.items() if 'decoder' in n], "lr": 1e-3},
                                {"params": [p for n, p in named.items() if 'decoder' not in n], "lr": 2e-3}], eps=1e-15)
        metrics = types.SimpleNamespace(total_loss=0.0, l2_loss=0.0, rgb_loss=0.0, num_samples=0)
        me = types.SimpleNamespace(pipeline=types.SimpleNamespace(nef=fr, zero_grad=fr.zero_grad), device='cpu', loss_lods=lods,
                                   train_dataset=types.SimpleNamespace(), tracker=types.SimpleNamespace(metrics=metrics), optimizer=opt)
        saved = getattr(C, "optim_step_groups")
        C.optim_step_groups = _torch_optim_groups
        try:
            tr = SDFTrainStep(fm, lr=1e-3, eps=1e-15, grid_lr_weight=2.0, optimizer='adam', only_last=only_last)
            for x, y in zip(X, Y):
                before = metrics.total_loss
                ref_step(me, {"coords": x, "sdf": y})
                loss = tr.step(x, y)
                assert abs(float(loss) * x.shape[0] - (metrics.total_loss - before)) <= 2e-5 * max(1.0, metrics.total_loss - before)
        finally:
            C.optim_step_groups = saved
        for (n1, p1), (n2, p2) in zip(fr.named_parameters(), fm.named_parameters()):
            np.testing.assert_allclose(p2.detach().numpy(), p1.detach().numpy(), rtol=1e-5, atol=2e-7, err_msg=n1)


@pytest.mark.parametrize("only_last", [True, False])
def test_dropin_sdf_trainer_class_equals_the_reference_methods(only_last):
    """wisp.trainers.SDFTrainer (what app/nglod constructs) through its own BaseTrainer.iterate() - next to a trainer assembled
    from the reference's method bodies compiled where they lie: BaseTrainer.iterate / begin_epoch / end_epoch / init_optimizer and
    SDFTrainer.pre_epoch / step (sdf_trainer.py:51-124).  512-free: 64 coordinates per batch, 3 epochs of 4 batches, Adam with the
    name-matched groups: the loss LODs chosen per epoch, per-step losses, the three metric sums and the parameters agree."""
    import time
    from wisp.datasets import SDFTensorDataset
    from wisp.trainers import SDFTrainer, ConfigSDFTrainer, ConfigAdam, ConfigDataloader

    class Field(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(8)
            self.grid = torch.nn.Module()
            self.grid.num_lods = 3
            self.grid.feats = torch.nn.Parameter(torch.randn(3, 32, 4) * 0.1)
            self.decoder = torch.nn.Linear(4 + 3, 1)

        def forward(self, coords=None, lod_idx=None, channels=None):
            cell = ((coords[:, 0] * 0.5 + 0.5) * 31).long().clamp(0, 31)
            f = self.grid.feats[: lod_idx + 1, cell].sum(0)
            out = self.decoder(torch.cat([f, coords], -1))
            return [out] if isinstance(channels, (list, tuple)) else out

    class Pipe(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.nef = Field()

    g = torch.Generator().manual_seed(2)
    X, Y = torch.rand(4 * 64, 3, generator=g) * 2 - 1, torch.randn(4 * 64, 1, generator=g) * 0.1

    class _Loader:                                   # fixed order: both sides see the same batches
        def __len__(self):
            return 4
        def __iter__(self):
            return iter([{"coords": X[i * 64:(i + 1) * 64], "sdf": Y[i * 64:(i + 1) * 64]} for i in range(4)])

    def make_cfg():
        return ConfigSDFTrainer(optimizer=ConfigAdam(lr=1e-3, eps=1e-15), dataloader=ConfigDataloader(batch_size=64), grid_lr_weight=2.0,
                                max_epochs=3, enable_amp=False, only_last=only_last)

    pipe_m = Pipe()
    tr = SDFTrainer(make_cfg(), pipe_m, SDFTensorDataset(X, Y), device='cpu')
    assert tr.iterations_per_epoch == 4                                    # 256 coordinates / 64 per batch
    tr.train_data_loader = _Loader()
    tr.is_optimization_running = True
    mine = []
    for _ in range(11):
        before = tr.tracker.metrics.total_loss if tr.iteration else 0.0
        tr.iterate()
        mine.append((tr.epoch, tr.iteration, list(tr.loss_lods), tr.tracker.metrics.total_loss, tr.tracker.metrics.l2_loss,
                     tr.tracker.metrics.num_samples))

    glb = dict(torch=_TorchWithoutNvtx(), time=time,
               instantiate=lambda cfg, params: torch.optim.Adam(params, lr=cfg.lr, eps=cfg.eps, weight_decay=cfg.weight_decay, betas=cfg.betas))
    class _Base:
        def pre_epoch(self_inner):
            me.pipeline.train()
            me.tracker.metrics.__dict__.update(total_loss=0.0, l2_loss=0.0, rgb_loss=0.0, num_samples=0)
    glb["super"] = lambda: _Base()
    body = {m: _reference_method("trainers/base_trainer.py", "BaseTrainer", m, glb)
            for m in ("iterate", "begin_epoch", "end_epoch", "is_first_iteration", "is_any_iterations_remaining", "reset_data_iterator",
                      "next_batch", "init_optimizer")}
    for m in ("pre_epoch", "step"):
        body[m] = _reference_method("trainers/sdf_trainer.py", "SDFTrainer", m, glb)
    noop = lambda self, *a, **k: None
    RefTrainer = type("RefTrainer", (), dict(
        body, pre_training=noop, post_training=noop, post_epoch=noop, post_step=noop, pre_step=noop, validate=noop,
        total_iterations=property(lambda self: (self.epoch - 1) * self.iterations_per_epoch + self.iteration),
        max_iterations=property(lambda self: self.max_epochs * self.iterations_per_epoch)))
    pipe_r = Pipe()
    me = RefTrainer()
    me.pipeline, me.device, me.cfg, me.enable_amp = pipe_r, 'cpu', make_cfg(), False
    me.tracker = types.SimpleNamespace(metrics=types.SimpleNamespace(total_loss=0.0, l2_loss=0.0, rgb_loss=0.0, num_samples=0),
                                       log_metric=lambda *a, **k: None)
    me.scene_state = types.SimpleNamespace(optimization=types.SimpleNamespace(elapsed_time=0.0, iterations_per_epoch=4))
    me.train_dataset = [0] * 256
    me.train_data_loader, me.train_data_loader_iter = _Loader(), None
    me.epoch, me.iteration, me.iterations_per_epoch, me.max_epochs = 1, 0, 4, 3
    me.is_optimization_running = True
    me.init_optimizer()
    me.train_dataset = types.SimpleNamespace()                              # (no sample_tex attribute: the plain branch)
    theirs = []
    for _ in range(11):
        me.iterate()
        theirs.append((me.epoch, me.iteration, list(me.loss_lods), me.tracker.metrics.total_loss, me.tracker.metrics.l2_loss,
                       me.tracker.metrics.num_samples))
    for a, b in zip(mine, theirs):
        assert a[:3] == b[:3] and a[5] == b[5], (a, b)
        np.testing.assert_allclose(a[3:5], b[3:5], rtol=2e-6)
    assert mine[0][2] == ([2] if only_last else [0, 1, 2])
    for (n1, p1), (n2, p2) in zip(pipe_r.nef.named_parameters(), pipe_m.nef.named_parameters()):
        assert n1 == n2
        np.testing.assert_allclose(p2.detach().numpy(), p1.detach().numpy(), rtol=1e-6, atol=1e-8, err_msg=n1)


@pytest.mark.parametrize("mode,steps", [("ray", 96), ("voxel", 6)])
def test_oracle_tracer_equals_the_reference_trace_body(mode, steps):
    """PackedRFTracer.trace (tracers/packed_rf_tracer.py:84-181), the method body compiled from the reference file, driven with
    the oracle's raymarch as `nef.grid.raymarch`, the oracle's radiance field as `nef(...)` and the oracle's restatement of the
    two Kaolin compositing leaves as `spc_render` - against oracle.nerf.trace, the tracer every GPU end-to-end test is compared
    with: the way the pieces are put together (optical thickness, exclusive transmittance, depth / alpha / hit / background
    scatter) is the reference's own code, bit for bit."""
    from oracle import nerf as onerf, raymarch as oray, render as orender
    from wisp.core import Rays, RenderBuffer
    spc_render = types.SimpleNamespace(exponential_integration=orender.exponential_integration, sum_reduce=orender.sum_reduce)
    trace = _reference_method("tracers/packed_rf_tracer.py", "PackedRFTracer", "trace",
                              dict(torch=_TorchWithoutNvtx(), spc_render=spc_render, RenderBuffer=RenderBuffer))
    rng = np.random.default_rng(60)
    blas = onerf.OracleBLAS.from_quantized_points(rng.integers(0, 16, size=(500, 3)), 4)
    res = [8, 16, 32, 64]
    torch.manual_seed(4)
    onef = onerf.OracleNeRF(res, 2, 10, 'cat', 0.3, 64, 1, True, 4)
    o = rng.normal(size=(120, 3)).astype(np.float32)
    o = 3.0 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = (-o + rng.normal(size=o.shape).astype(np.float32) * 0.4)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    O, D = torch.from_numpy(o), torch.from_numpy(d)
    if mode == "ray":
        jit = rng.uniform(size=(120, steps)).astype(np.float32)
    else:
        from oracle import spc as ospc
        nn = ospc.raytrace(blas.octree, blas.points, blas.pyramid, blas.exsum, o, d, blas.max_level, True)[0].shape[0]
        jit = rng.uniform(size=(nn, steps)).astype(np.float32)
    bg = (0.2, 0.5, 0.7)
    with torch.no_grad():
        want = onerf.trace(onef, blas, O, D, 1.0, 5.0, steps, jit, bg, mode, with_depth=True)

    class Grid:
        num_lods, active_lods = len(res), [blas.max_level] * len(res)

        @staticmethod
        def raymarch(rays, level, num_samples, raymarch_type):
            rm = want["raymarch"]                             # the oracle's samples for exactly these rays
            assert raymarch_type == mode and num_samples == steps
            return types.SimpleNamespace(ridx=torch.from_numpy(rm["ridx"]), samples=torch.from_numpy(rm["samples"]),
                                         deltas=torch.from_numpy(rm["deltas"]), depth_samples=torch.from_numpy(rm["depth_samples"]),
                                         boundary=torch.from_numpy(rm["boundary"]), pack_info=None)

    class Nef:
        grid = Grid()

        def __call__(self, coords=None, ray_d=None, lod_idx=None, channels=None):
            out = onef.rgba(coords, ray_d, lod_idx)
            return [out[c] for c in channels] if isinstance(channels, (list, tuple)) else out[channels]

    me = types.SimpleNamespace(bg_color=torch.tensor(bg), prev_num_samples=None)
    with torch.no_grad():
        rb = trace(me, Nef(), Rays(O, D, dist_min=1.0, dist_max=5.0), {"rgb", "depth", "alpha", "hit"}, set(), lod_idx=None,
                   raymarch_type=mode, num_steps=steps, bg_color=bg)
    assert me.prev_num_samples == want["raymarch"]["ridx"].shape[0] > 500
    assert torch.equal(rb.rgb, want["rgb"]) and torch.equal(rb.alpha, want["alpha"])
    assert torch.equal(rb.depth, want["depth"]) and torch.equal(rb.hit, want["hit"])
    assert int(rb.hit.sum()) > 20


def _reference_function(rel, fn_name, glb):
    """compile ONE module-level function of a reference file (decorators dropped)"""
    import ast
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read(), path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == fn_name)
    fn.decorator_list = []
    ns = dict(glb)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns[fn_name]


class _TorchWithDraws(_TorchWithoutNvtx):
    """... and torch.rand / torch.rand_like hand out prepared jitter instead of fresh draws"""
    def __init__(self, draws):
        super().__init__()
        self._draws = draws

    def rand(self, *shape, **kw):
        assert tuple(shape) == tuple(self._draws.shape), (shape, self._draws.shape)
        return self._draws.clone()

    def rand_like(self, t, **kw):
        assert tuple(t.shape) == tuple(self._draws.shape), (t.shape, self._draws.shape)
        return self._draws.clone()


@pytest.mark.parametrize("mode,steps", [("ray", 128), ("voxel", 5), ("uniform", 48)])
def test_oracle_raymarch_equals_the_reference_method_bodies(mode, steps):
    """OctreeAS._raymarch_ray / _raymarch_voxel / _raymarch_uniform (accelstructs/octree_as.py:188-374) and the helper
    fast_filter_method, compiled from the reference file and run with the oracle's restatements of the Kaolin leaves
    (query, raytrace, pack boundaries, inclusive sum) and of the uniform-sample kernel (itself pinned to the reference's CUDA
    body, test_oracle_golden) plugged in, and the random draws injected - against oracle.raymarch.*, what the HIP raymarch
    kernels are compared with bit for bit: ray indices, sample positions, depths, deltas and pack boundaries."""
    from typing import Tuple
    from oracle import nerf as onerf, raymarch as oray, spc as ospc
    from wisp.accelstructs import ASRaymarchResults
    from wisp.core import Rays
    rng = np.random.default_rng(70)
    blas = onerf.OracleBLAS.from_quantized_points(rng.integers(0, 32, size=(1500, 3)), 5)
    o = rng.normal(size=(150, 3)).astype(np.float32)
    o = 3.0 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = (-o + rng.normal(size=o.shape).astype(np.float32) * 0.5)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = Rays(torch.from_numpy(o), torch.from_numpy(d), dist_min=1.0, dist_max=5.0)
    level = blas.max_level
    nug = ospc.raytrace(blas.octree, blas.points, blas.pyramid, blas.exsum, o, d, level, True)
    if mode == "ray":
        jit = rng.uniform(size=(150, steps)).astype(np.float32)
        want = oray.raymarch_ray(blas.octree, blas.exsum, o, d, 1.0, 5.0, steps, level, jit)
    elif mode == "voxel":
        jit = rng.uniform(size=(nug[0].shape[0], steps)).astype(np.float32)
        want = oray.raymarch_voxel(blas.octree, blas.points, blas.pyramid, blas.exsum, o, d, steps, level, jit)
    else:
        jit = np.zeros((1, 1), np.float32)
        want = oray.raymarch_uniform(blas.octree, blas.points, blas.pyramid, blas.exsum, o, d, steps, level)
    tproxy = _TorchWithDraws(torch.from_numpy(jit))
    sampling = _exec_reference("ops/spc/sampling.py")
    sampling["torch"] = tproxy                                   # its rand_like -> the injected jitter
    wisp_spc_ops = types.SimpleNamespace(sample_from_depth_intervals=sampling["sample_from_depth_intervals"],
                                         expand_pack_boundary=sampling["expand_pack_boundary"])
    first_hit = lambda ridx: torch.from_numpy(ospc.mark_pack_boundaries(ridx.numpy()))
    spc_render = types.SimpleNamespace(mark_pack_boundaries=first_hit, mark_first_hit=first_hit)
    kaolin_C = types.SimpleNamespace(render=types.SimpleNamespace(spc=types.SimpleNamespace(
        inclusive_sum_cuda=lambda x: torch.from_numpy(ospc.inclusive_sum(x.numpy())))))

    def uniform_sample_cuda(scale, ridx, depth, insum):
        out = oray.uniform_sample(scale, ridx.numpy(), depth.numpy(), insum.numpy())
        return [torch.from_numpy(out["ridx"]), torch.from_numpy(out["depth_samples"]), torch.from_numpy(out["boundary"])]
    wisp_C = types.SimpleNamespace(ops=types.SimpleNamespace(uniform_sample_cuda=uniform_sample_cuda))
    glb = dict(torch=tproxy, np=np, Tuple=Tuple, ASRaymarchResults=ASRaymarchResults, wisp_spc_ops=wisp_spc_ops,
               spc_render=spc_render, _C=kaolin_C, wisp_C=wisp_C)
    glb["fast_filter_method"] = _reference_function("accelstructs/octree_as.py", "fast_filter_method", glb)
    method = _reference_method("accelstructs/octree_as.py", "OctreeAS", "_raymarch_" + mode, glb)
    me = types.SimpleNamespace(
        max_level=level,
        query=lambda coords, level=None, with_parents=False: types.SimpleNamespace(
            pidx=torch.from_numpy(ospc.query(blas.octree, blas.exsum, coords.numpy(), level))),
        raytrace=lambda rays, level=None, with_exit=False: types.SimpleNamespace(
            ridx=torch.from_numpy(nug[0]).int(), pidx=torch.from_numpy(nug[1]).int(), depth=torch.from_numpy(nug[2])))
    res = method(me, rays, steps, level)
    assert res.ridx.shape[0] == want["ridx"].shape[0] > 150
    assert np.array_equal(res.ridx.numpy(), want["ridx"])                               # the same samples survive, in the same order
    assert np.array_equal(res.boundary.numpy().astype(bool), want["boundary"])
    # values: the oracle restates the arithmetic as the GPU evaluates it (product of addcmul rounded before the add);
    # torch's CPU kernels fuse / order a few of these operations differently - a last-bit matter
    np.testing.assert_allclose(res.depth_samples.numpy(), want["depth_samples"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(res.deltas.numpy(), want["deltas"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(res.samples.numpy(), want["samples"], rtol=0, atol=1e-6)


def test_oracle_prune_equals_the_reference_method_body(monkeypatch):
    """NeuralRadianceField.prune (models/nefs/nerf.py:175-212), the method body compiled from the reference file with the
    random draws injected and the oracle's field / octree builder plugged in - against oracle.nerf.prune: decayed occupancy,
    density query at the jittered cell positions, running maximum, threshold, rebuilt octree."""
    from oracle import nerf as onerf
    rng = np.random.default_rng(80)
    blas = onerf.OracleBLAS.make_dense(4)
    dense = blas.level_points().copy()
    cells = dense.shape[0]
    torch.manual_seed(6)
    onef = onerf.OracleNeRF([8, 16, 32, 64], 2, 10, 'cat', 0.5, 64, 1, True, 4)
    unit = torch.from_numpy(rng.uniform(size=(cells, 3)).astype(np.float32))
    views = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(cells, 3)).astype(np.float32)), dim=1)
    occ0 = torch.from_numpy(rng.uniform(0, 2, cells).astype(np.float32))
    with torch.no_grad():
        dens = onef.rgba((((torch.from_numpy(dense.astype(np.float32)) + unit) / 16) * 2 - 1), views)["density"][:, 0]
    thr = float(torch.maximum(dens, occ0 * 0.95).median())                          # keeps about half of the cells
    want_blas, want_occ = onerf.prune(onef, blas, occ0.clone(), dense, 0.95, thr, unit, views)

    class Grid:
        pass

    class Blas:
        max_level = blas.max_level
        rebuilt = None

        @classmethod
        def from_quantized_points(cls, pts, level):
            b = cls()
            b.oracle = onerf.OracleBLAS.from_quantized_points(pts.numpy(), level)
            return b
    grid = Grid()
    grid.occupancy, grid.dense_points, grid.blas = occ0.clone(), torch.from_numpy(dense.astype(np.int16)), Blas()
    me = types.SimpleNamespace(prune_density_decay=0.95, prune_min_density=thr, grid=grid,
                               forward=lambda coords=None, ray_d=None, channels=None: onef.rgba(coords, ray_d)[channels])
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)            # `.cuda()` calls of the body on a CPU box
    body = _reference_method("models/nefs/nerf.py", "NeuralRadianceField", "prune",
                             dict(torch=_TorchWithDraws(unit), HashGrid=Grid, TriplanarGrid=Grid,
                                  sample_unif_sphere=lambda n: views.numpy()))
    body(me)
    assert torch.equal(grid.occupancy, want_occ)
    assert want_blas is not None and np.array_equal(grid.blas.oracle.octree, want_blas.octree)
    kept = int(grid.blas.oracle.pyramid[0, blas.max_level])
    assert 0.3 * cells < kept < 0.7 * cells


@pytest.mark.parametrize("multiscale", ["cat", "sum"])
def test_oracle_hash_grid_equals_the_whole_reference_stack_on_the_host(multiscale):
    """The reference's complete hash-grid path on the CPU: `HashGrid.interpolate` (models/grids/hash_grid.py:205-233, body
    compiled from the file) -> `wisp/ops/grid.py` (hashgrid + the HashGridInterpolate autograd function, executed in place) ->
    `wisp._C.ops.hashgrid_interpolate_cuda / _backward_cuda` = the reference's own kernel bodies compiled for the host
    (oracle/_ref, oracle/build_ref.sh) - against oracle.hashgrid.grid_interpolate, forward (bit-exact, every lod_idx, [B,3]
    and [B,S,3] inputs) and the gradient w.r.t. the table."""
    from oracle import hashgrid as ohash, ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built")
    from wisp.models.grids.utils import MultiTable

    def fwd(coords, codebook, first_idx, resolution, bitwidth):
        res = [int(r) for r in resolution.reshape(-1).tolist()]
        return torch.from_numpy(ref_lib.hashgrid_forward(coords.numpy(), codebook.detach().numpy(), first_idx.numpy(), res, int(bitwidth)))

    def bwd(coords, grad_output, codebook, first_idx, resolution, bitwidth, feature_dim, require_grad_coords):
        res = [int(r) for r in resolution.reshape(-1).tolist()]
        g = ref_lib.hashgrid_backward(coords.numpy(), grad_output.numpy(), codebook.detach().numpy(), first_idx.numpy(), res, int(bitwidth))
        return [torch.empty(0), torch.from_numpy(g)]
    native = types.ModuleType("wisp._C")
    native.ops = types.SimpleNamespace(hashgrid_interpolate_cuda=fwd, hashgrid_interpolate_backward_cuda=bwd)
    import wisp
    stubs = dict(_kaolin_stub())
    stubs["wisp._C"] = native
    saved = {k: sys.modules.get(k) for k in stubs}
    saved_attr = getattr(wisp, "_C", None)
    sys.modules.update(stubs)
    wisp._C = native
    try:
        grid_ops = _exec_reference("ops/grid.py")                 # `import wisp._C as wisp_C` -> the host build of the reference kernels
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        if saved_attr is not None:
            wisp._C = saved_attr
    interpolate = _reference_method("models/grids/hash_grid.py", "HashGrid", "interpolate",
                                    dict(torch=torch, grid_ops=types.SimpleNamespace(hashgrid=grid_ops["hashgrid"])))
    res, bw, F = [4, 6, 8, 10, 13, 16, 23, 32], 10, 2
    torch.manual_seed(7)
    table = MultiTable(res, 3, F, 0.3, 2 ** bw)
    me = types.SimpleNamespace(codebook=table, codebook_bitwidth=bw, multiscale_type=multiscale, feature_dim=F, resolutions=res)
    rng = np.random.default_rng(90)
    flat = torch.from_numpy(rng.uniform(-1, 1, (120, 3)).astype(np.float32))
    for coords in (flat, flat.reshape(30, 4, 3)):
        for lod_idx in (0, 3, len(res) - 1):
            got = interpolate(me, coords, lod_idx)
            want = ohash.grid_interpolate(coords, lod_idx, multiscale, F, res, bw, table.feats.detach(), table.begin_idxes)
            assert got.shape == want.shape and torch.equal(got.detach(), want), (tuple(coords.shape), lod_idx)
    # backward through the reference's autograd function and kernels vs the oracle's
    go = torch.from_numpy(rng.normal(size=(120, F * len(res) if multiscale == "cat" else F)).astype(np.float32))
    table.feats.grad = None
    interpolate(me, flat, len(res) - 1).backward(go)
    ref_grad = table.feats.grad.clone()
    t2 = table.feats.detach().clone().requires_grad_(True)
    ohash.grid_interpolate(flat, len(res) - 1, multiscale, F, res, bw, t2, table.begin_idxes).backward(go)
    np.testing.assert_allclose(t2.grad.numpy(), ref_grad.numpy(), rtol=0, atol=2e-5)      # the reference adds sequentially in fp32
    assert float(ref_grad.abs().max()) > 0.1


@pytest.mark.parametrize("num_steps,step_size,min_dis", [(48, 0.8, 3e-4), (6, 1.0, 1e-4)])
def test_oracle_sphere_tracer_equals_the_reference_trace_body(num_steps, step_size, min_dis):
    """PackedSDFTracer.trace (tracers/packed_sdf_tracer.py:57-174): the method body compiled from the reference file, with the
    reference's own find_depth_bound wrapper (ops/geometric.py:15-22) over its kernel body built for the host and its own
    finitediff_gradient (ops/differential/gradients.py:29-45) - driven with the oracle's octree raytrace and an analytic distance
    function - against oracle.sdf.sphere_trace, the tracer the GPU SDF tests are compared with.  Marching order, both convergence
    tests, the far plane, the nugget jump and the output scatter are the reference's code; every buffer must be identical."""
    import torch.nn.functional as F
    from oracle import nerf as onerf, sdf as osdf, spc as ospc, ref_lib
    from wisp.core import Rays, RenderBuffer

    def sdf_fn(x):
        return (x.norm(dim=-1, keepdim=True) - 0.55) + 0.02 * torch.sin(9.0 * x[..., 0:1]) * torch.cos(7.0 * x[..., 1:2])

    level = 5
    g = (np.stack(np.meshgrid(*[np.arange(32)] * 3, indexing="ij"), -1).reshape(-1, 3) + 0.5) / 16.0 - 1.0
    shell = np.abs(np.linalg.norm(g, axis=1) - 0.55) < 0.09
    blas = onerf.OracleBLAS.from_quantized_points(((g[shell] + 1.0) * 16.0).astype(np.int64), level)
    rng = np.random.default_rng(77)
    o = rng.normal(size=(160, 3)).astype(np.float32)
    o = (2.5 * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    d = -o + rng.normal(size=o.shape).astype(np.float32) * 0.55          # some rays graze or miss the shell
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    O, D = torch.from_numpy(o), torch.from_numpy(d)
    dist_max = 6.0
    want = osdf.sphere_trace(sdf_fn, blas, O, D, dist_max, level, num_steps, step_size, min_dis)

    kernel = ref_lib.find_depth_bound if ref_lib.available() else osdf.find_depth_bound       # the latter is golden-pinned
    c_ext = types.SimpleNamespace(render=types.SimpleNamespace(find_depth_bound_cuda=lambda q, cur, dep: torch.from_numpy(
        kernel(q.numpy(), cur.numpy(), dep.numpy()))))
    find_depth_bound = _reference_function("ops/geometric.py", "find_depth_bound", dict(torch=torch, _C=c_ext))
    finitediff = _reference_function("ops/differential/gradients.py", "finitediff_gradient", dict(torch=torch))
    spc_render = types.SimpleNamespace(mark_pack_boundaries=lambda r: torch.from_numpy(ospc.mark_pack_boundaries(r.numpy())))
    trace = _reference_method("tracers/packed_sdf_tracer.py", "PackedSDFTracer", "trace",
                              dict(torch=_TorchWithoutNvtx(), F=F, spc_render=spc_render, RenderBuffer=RenderBuffer,
                                   find_depth_bound=find_depth_bound, finitediff_gradient=finitediff))
    seen = {}

    class Grid:
        num_lods, active_lods = 1, [level]

        @staticmethod
        def raytrace(rays, lvl, with_exit=False):
            assert lvl == level and with_exit
            ridx, pidx, depth = ospc.raytrace(blas.octree, blas.points, blas.pyramid, blas.exsum, rays.origins.numpy(),
                                              rays.dirs.numpy(), lvl, with_exit=True)
            seen["nuggets"] = ridx.shape[0]
            return types.SimpleNamespace(ridx=torch.from_numpy(ridx), pidx=torch.from_numpy(pidx), depth=torch.from_numpy(depth.copy()))

    class Nef:
        grid = Grid()

        def __call__(self, coords=None, lod_idx=None, pidx=None, channels=None):
            assert channels == "sdf" and lod_idx == 0 and pidx.shape[0] == coords.shape[0]
            return sdf_fn(coords)

        @staticmethod
        def get_forward_function(channel):
            assert channel == "sdf"
            return sdf_fn

    rb = trace(types.SimpleNamespace(), Nef(), Rays(O, D, dist_min=0.0, dist_max=dist_max), {"rgb", "normal", "depth", "hit"}, set(),
               lod_idx=None, num_steps=num_steps, step_size=step_size, min_dis=min_dis)
    assert seen["nuggets"] > 1000
    for name in ("xyz", "depth", "hit", "normal", "rgb", "alpha"):
        assert torch.equal(getattr(rb, name), want[name]), name
    hits = int(rb.hit.sum())
    assert (20 < hits < 160) if num_steps == 48 else hits < 100           # the short run stops with rays still marching


def _octree_grid_scene(num_lods=3, base_lod=2, feature_dim=4, codes=None, seed=8):
    """a sparse level-(base_lod + num_lods - 1) tree with its dual corners, one feature (or logits) table per active level"""
    from oracle import nerf as onerf, spc as ospc
    rng = np.random.default_rng(seed)
    top = base_lod + num_lods - 1
    blas = onerf.OracleBLAS.from_quantized_points(rng.integers(0, 2 ** top, size=(90, 3)), top)
    pd, pyd = ospc.make_dual(blas.points, blas.pyramid)
    trinkets, _ = ospc.make_trinkets(blas.points, blas.pyramid, pd, pyd)
    active = list(range(base_lod, top + 1))
    torch.manual_seed(seed)
    width = feature_dim if codes is None else codes
    feats = [torch.randn(int(pyd[0, l]) + 1, width, requires_grad=True) for l in active]
    x = rng.uniform(-1.05, 1.05, (300, 3)).astype(np.float32)                  # a few land outside the cube -> pidx -1
    return blas, torch.from_numpy(np.asarray(trinkets)), active, feats, torch.from_numpy(x)


def _reference_octree_grid(blas, trinkets, active, feats, base_lod, multiscale, feature_dim, glb, cls_file=None, extra=None):
    """`self` for the compiled reference methods: the attributes OctreeGrid.interpolate / _interpolate read, the oracle's octree
    query behind `blas.query`, and the Kaolin leaves restated by the oracle as `spc_ops`."""
    import functools
    from oracle import octree_grid as og, spc as ospc

    class Blas:
        points = torch.from_numpy(np.asarray(blas.points))
        pyramid = torch.from_numpy(np.asarray(blas.pyramid))

        @staticmethod
        def query(coords, level=None, with_parents=False):
            return types.SimpleNamespace(pidx=torch.from_numpy(ospc.query(blas.octree, blas.exsum, coords.detach().numpy(), level,
                                                                          with_parents=with_parents)))

    spc_ops = types.SimpleNamespace(
        unbatched_interpolate_trilinear=lambda c, pidx, pts, tr, f, lod: og.interpolate_trilinear(c, pidx.long(), pts, tr, f.float(), lod,
                                                                                                   half_round=True),
        coords_to_trilinear_coeffs=lambda c, pts, lod: og.trilinear_coeffs(c, pts.long(), lod))
    glb = dict(glb, torch=torch, spc_ops=spc_ops)
    me = types.SimpleNamespace(blas=Blas, trinkets=trinkets, active_lods=active, base_lod=base_lod, features=feats,
                               interpolation_type='linear', multiscale_type=multiscale, feature_dim=feature_dim, training=True,
                               **(extra or {}))
    interp_cls = ("models/grids/octree_grid.py", "OctreeGrid") if cls_file is None else cls_file
    me._interpolate = functools.partial(_reference_method(*interp_cls, "_interpolate", glb), me)
    if cls_file is not None:
        me._index_features = functools.partial(_reference_method(*cls_file, "_index_features", glb), me)
    me.interpolate = functools.partial(_reference_method("models/grids/octree_grid.py", "OctreeGrid", "interpolate", glb), me)
    return me


@pytest.mark.parametrize("multiscale", ["cat", "sum"])
def test_oracle_octree_grid_equals_the_reference_interpolate_bodies(multiscale):
    """OctreeGrid.interpolate + _interpolate (models/grids/octree_grid.py:130-219) compiled from the reference file over the oracle's
    octree query and its restatement of kaolin's unbatched_interpolate_trilinear - against oracle.octree_grid.octree_grid_interpolate,
    what the HIP octree-grid kernels are compared with: which pidx column feeds which level (`pidx[..., base_lod:]`, `_interpolate(...,
    i)` -> active_lods[i]), the lod_idx == 0 shortcut, 'cat' / 'sum', output shapes, cells outside the tree, gradients per table."""
    from oracle import octree_grid as og
    base_lod, F = 2, 4
    blas, trinkets, active, feats, x = _octree_grid_scene(3, base_lod, F)
    me = _reference_octree_grid(blas, trinkets, active, feats, base_lod, multiscale, F, {})
    for lod_idx in (0, 1, 2):
        for coords in (x, x.reshape(60, 5, 3)):
            if lod_idx == 0 and coords.ndim == 3:
                continue              # the reference reuses sample 0's cell for the whole row there (base-level samples share a cell)
            for f in feats:
                f.grad = None
            got = me.interpolate(coords, lod_idx)
            got.square().sum().backward()
            g_ref = [None if f.grad is None else f.grad.clone() for f in feats]
            for f in feats:
                f.grad = None
            want = og.octree_grid_interpolate(blas, trinkets, feats, coords, lod_idx, base_lod, active, multiscale, F, half_round=True)
            want.square().sum().backward()
            width = F if (multiscale == 'sum' or lod_idx == 0) else F * (lod_idx + 1)
            assert got.shape == (*coords.shape[:-1], width)
            assert torch.equal(got.reshape(-1, width), want), (lod_idx, coords.shape)
            assert int((got.reshape(-1, width).abs().sum(-1) == 0).sum()) >= 5                # the samples outside the cube
            for i, f in enumerate(feats):
                assert (f.grad is None) == (g_ref[i] is None) == (i > lod_idx)
                if f.grad is not None:
                    assert torch.equal(f.grad, g_ref[i]) and float(f.grad.abs().max()) > 0


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("multiscale", ["cat", "sum"])
def test_oracle_codebook_grid_equals_the_reference_interpolate_bodies(multiscale, training):
    """CodebookOctreeGrid._index_features + _interpolate (models/grids/codebook_grid.py:103-172) under the inherited
    OctreeGrid.interpolate, all compiled from the reference files - against oracle.octree_grid.codebook_grid_interpolate: straight-through
    softmax keys while training / argmax rows in eval, invalid cells, level selection, 'cat' / 'sum', gradients into logits and
    dictionaries."""
    import torch.nn.functional as F_
    from oracle import octree_grid as og
    base_lod, F, K = 2, 4, 16
    blas, trinkets, active, logits, x = _octree_grid_scene(3, base_lod, F, codes=K)
    torch.manual_seed(21)
    dictionary = [torch.randn(K, F, requires_grad=True) for _ in active]
    me = _reference_octree_grid(blas, trinkets, active, logits, base_lod, multiscale, F, dict(F=F_),
                                cls_file=("models/grids/codebook_grid.py", "CodebookOctreeGrid"), extra=dict(dictionary=dictionary))
    me.training = training
    for lod_idx in (0, 2):
        leaves = logits + dictionary
        for t in leaves:
            t.grad = None
        got = me.interpolate(x, lod_idx)
        got.square().sum().backward()
        g_ref = [None if t.grad is None else t.grad.clone() for t in leaves]
        for t in leaves:
            t.grad = None
        want = og.codebook_grid_interpolate(blas, trinkets, logits, dictionary, x, lod_idx, active, multiscale, F, training)
        want.square().sum().backward()
        assert got.shape == want.shape and torch.allclose(got, want, atol=1e-6, rtol=0), float((got - want).abs().max())
        assert int((got.abs().sum(-1) == 0).sum()) >= 5
        for t, g in zip(leaves, g_ref):
            assert (t.grad is None) == (g is None)
            if g is not None:
                assert torch.allclose(t.grad, g, atol=1e-5, rtol=1e-5)
        assert dictionary[0].grad is not None and (logits[0].grad is not None) == training   # eval: argmax rows, no logits gradient


@pytest.mark.parametrize("position_input", [True, False])
def test_oracle_sdf_field_composition_equals_the_reference_sdf_body(position_input):
    """NeuralSDF.sdf (models/nefs/neural_sdf.py:120-155) compiled from the reference file, with its own init_embedder choice
    (Identity for 'none' + position_input, :89-100), over the oracle's octree-grid lookup and decoder - against the composition the
    GPU SDF tests use as the oracle field, decoder(cat([position, features])): concatenation order, [B,3] and [B,S,3] shapes, the
    default lod_idx, the empty batch."""
    from oracle import nerf as onerf, octree_grid as og
    base_lod, F = 2, 4
    blas, trinkets, active, feats, x = _octree_grid_scene(3, base_lod, F)
    torch.manual_seed(31)
    embed_dim = 3 if position_input else 0
    dec = onerf.OracleDecoder(F + embed_dim, 1, 32, 1, True)

    def lookup(coords, lod_idx):
        out = og.octree_grid_interpolate(blas, trinkets, feats, coords, lod_idx, base_lod, active, 'sum', F, half_round=True)
        return out.reshape(*coords.shape[:-1], F)

    init_embedder = _reference_method("models/nefs/neural_sdf.py", "NeuralSDF", "init_embedder", dict(torch=torch))
    embedder, dim = init_embedder(None, 'none', None, position_input)
    assert dim == embed_dim and (embedder is None) == (not position_input)
    me = types.SimpleNamespace(grid=types.SimpleNamespace(num_lods=3, interpolate=lookup), pos_embedder=embedder, pos_embed_dim=dim,
                               decoder=dec)
    sdf = _reference_method("models/nefs/neural_sdf.py", "NeuralSDF", "sdf", dict(torch=torch))
    for coords, lod_idx in ((x, 1), (x, None), (x.reshape(60, 5, 3), 2)):
        got = sdf(me, coords, lod_idx)["sdf"]
        f = lookup(coords, 2 if lod_idx is None else lod_idx)
        want = dec(torch.cat([coords, f], -1) if position_input else f)
        assert got.shape == (*coords.shape[:-1], 1) and torch.equal(got, want)
    assert float(got.detach().abs().max()) > 0
    empty = sdf(me, torch.zeros(0, 3), None)["sdf"]
    assert empty.shape == (0, 1)


@pytest.mark.parametrize("lod_idx", [None, 9])
def test_oracle_radiance_field_equals_the_reference_rgba_body(lod_idx):
    """NeuralRadianceField.rgba (models/nefs/nerf.py:219-264), the method body compiled from the reference file, with the reference's
    own PositionalEmbedder as view embedder and the oracle's hash-grid lookup as `grid.interpolate` - against OracleNeRF.rgba, the field
    every GPU NeRF test is compared with: default lod_idx, the [batch, effective_feature_dim] reshape, density features -> view embedding
    concatenation, colour decoded from fdir[..., 1:], relu density."""
    from oracle import hashgrid as ohash, nerf as onerf
    emb_cls = _exec_reference("models/embedders/positional_embedder.py")["PositionalEmbedder"]
    res = [16, 32, 64, 128, 256, 300, 350, 400, 420, 440, 460, 470, 480, 490, 500, 512]
    torch.manual_seed(12)
    onef = onerf.OracleNeRF(res, 2, 10, 'cat', 0.1, 64, 1, True, 4)
    emb = emb_cls(4, 3, log_sampling=True, include_input=True, input_dim=3)
    assert emb.out_dim == onef.view_embed_dim
    seen = []

    class Grid:
        active_lods, multiscale_type, feature_dim, num_lods = list(range(len(res))), 'cat', 2, len(res)

        @staticmethod
        def interpolate(coords, lod):
            seen.append(lod)
            return ohash.grid_interpolate(coords, lod, 'cat', 2, res, 10, onef.grid.codebook.feats, onef.begin_idxes)

    me = types.SimpleNamespace(grid=Grid, pos_embedder=None, view_embedder=emb, view_embedder_type='positional',
                               view_embed_dim=emb.out_dim, decoder_density=onef.decoder_density, decoder_color=onef.decoder_color)
    me.effective_feature_dim = lambda: _reference_method("models/nefs/nerf.py", "NeuralRadianceField", "effective_feature_dim", {})(me)
    rgba = _reference_method("models/nefs/nerf.py", "NeuralRadianceField", "rgba", dict(torch=torch))
    rng = np.random.default_rng(13)
    coords = torch.from_numpy(rng.uniform(-1, 1, (300, 3)).astype(np.float32))
    dirs = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(300, 3)).astype(np.float32)), dim=1)
    with torch.no_grad():
        got = rgba(me, coords, dirs, lod_idx)
        want = onef.rgba(coords, dirs, lod_idx)
    assert seen == [len(res) - 1 if lod_idx is None else lod_idx]
    assert got["rgb"].shape == (300, 3) and got["density"].shape == (300, 1)
    assert torch.equal(got["rgb"], want["rgb"]) and torch.equal(got["density"], want["density"])
    assert float(got["density"].max()) > 0 and float(got["rgb"].std()) > 0


@pytest.mark.parametrize("ortho", [False, True])
def test_oracle_ray_generation_equals_the_reference_function_bodies(ortho):
    """generate_default_grid / generate_centered_pixel_coords / _to_ndc_coords / generate_pinhole_rays / generate_ortho_rays
    (ops/raygen/raygen.py:16-119) compiled from the reference file and run on an adapter around this package's LookAtCamera (kaolin's
    Camera surface: width, height, x0, y0, tan_half_fov(CameraFOV), fov_distance, extrinsics.inv_transform_rays - the last restated as
    R^T (x - t), Kaolin's leaf) - against oracle.raygen, what the wisp_generate_rays kernel is compared with on the GPU: pixel centres,
    principal-point signs (x - x0, y + y0), NDC scaling, the flipped y axis, -z viewing direction, ortho plane scaling, normalisation."""
    from oracle import raygen as oray
    from wisp.core import Rays
    from wisp.ops.raygen import LookAtCamera
    W, H = 40, 24
    cam = LookAtCamera(eye=(1.5, 0.8, 2.5), at=(0.1, -0.2, 0.0), up=(0, 1, 0), fov=0.6911112, width=W, height=H, near=0.5, far=7.0,
                       x0=1.75, y0=-0.6, fov_distance=1.3)
    m = cam.view_matrix()[0]
    R, t = m[:3, :3], m[:3, 3]
    fov_axis = types.SimpleNamespace(HORIZONTAL='horizontal', VERTICAL='vertical')

    class Extrinsics:
        @staticmethod
        def inv_transform_rays(orig, dirs):
            return ((orig - t) @ R)[None], (dirs @ R)[None]                  # rows: R^T (x - t), R^T d

    kcam = types.SimpleNamespace(device=torch.device('cpu'), dtype=torch.float32, width=W, height=H, x0=cam.x0, y0=cam.y0,
                                 near=cam.near, far=cam.far, fov_distance=cam.fov_distance, extrinsics=Extrinsics,
                                 tan_half_fov=lambda axis: cam.tan_half_fov(axis))
    glb = dict(torch=torch, Rays=Rays, CameraFOV=fov_axis, Camera=object)
    glb["generate_default_grid"] = _reference_function("ops/raygen/raygen.py", "generate_default_grid", glb)
    glb["_to_ndc_coords"] = _reference_function("ops/raygen/raygen.py", "_to_ndc_coords", glb)
    grid = _reference_function("ops/raygen/raygen.py", "generate_centered_pixel_coords", glb)
    gen = _reference_function("ops/raygen/raygen.py", "generate_ortho_rays" if ortho else "generate_pinhole_rays", glb)
    for res in ((None, None), (20, 12)):                                      # native grid, and a coarser one scaled to the image
        py, px = grid(W, H, *res) if res[0] else grid(W, H, W, H)
        opy, opx = oray.centered_pixel_coords(W, H, *res)
        assert np.array_equal(py.numpy(), opy) and np.array_equal(px.numpy(), opx)
        rays = gen(kcam, (py, px))
        if ortho:
            sx, sy, x0, y0 = np.float32(cam.fov_distance) * np.float32(W / H), cam.fov_distance, 0.0, 0.0
        else:
            sx, sy, x0, y0 = cam.tan_half_fov('horizontal'), cam.tan_half_fov('vertical'), cam.x0, cam.y0
        o, d = oray.generate_rays(opx, opy, ortho, x0, y0, W, H, sx, sy, R.numpy(), t.numpy())
        assert rays.origins.shape == (py.numel(), 3) and (rays.dist_min, rays.dist_max) == (0.5, 7.0)
        np.testing.assert_allclose(rays.origins.numpy(), o, atol=2e-6, rtol=0)
        np.testing.assert_allclose(rays.dirs.numpy(), d, atol=2e-6, rtol=0)
    assert float(np.abs(d - d[0]).max()) == 0.0 if ortho else float(np.abs(d - d[0]).max()) > 0.1


def _kaolin_spc_leaves():
    """kaolin.ops.spc as the oracle restates it (torch in, torch out) - the leaves under the reference's SPC builders."""
    from oracle import spc as ospc
    t = torch.from_numpy

    def scan_octrees(octree, lengths):
        lvl, pyramid, exsum = ospc.scan_octree(octree.numpy())
        return torch.tensor([lvl]), t(np.asarray(pyramid))[None], t(np.asarray(exsum))

    def make_trinkets(points, pyramid, pd, pyd):
        tr, par = ospc.make_trinkets(points.numpy(), pyramid.numpy(), pd.numpy(), pyd.numpy())
        return t(np.asarray(tr)), t(np.asarray(par))

    return types.SimpleNamespace(
        quantize_points=lambda x, level: t(ospc.quantize_points(x.numpy(), level)),
        points_to_morton=lambda p: t(ospc.points_to_morton(p.numpy())),
        morton_to_points=lambda m: t(ospc.morton_to_points(m.numpy())),
        unbatched_points_to_octree=lambda p, level, sorted=False: t(ospc.points_to_octree(p.numpy(), level)),
        scan_octrees=scan_octrees,
        generate_points=lambda octree, pyramid, exsum: t(ospc.generate_points(octree.numpy(), pyramid[0].numpy(), exsum.numpy())),
        unbatched_make_dual=lambda p, pyr: tuple(t(np.asarray(a)) for a in ospc.make_dual(p.numpy(), pyr.numpy())),
        unbatched_make_trinkets=make_trinkets)


def test_oracle_and_package_spc_builders_equal_the_reference_function_bodies(monkeypatch):
    """create_dense_octree, make_trilinear_spc (ops/spc/constructors.py:14-47), pointcloud_to_octree, octree_to_spc
    (conversions.py:15-48,72-88) and dilate_points (processing.py:13-47) compiled from the reference files over the oracle's restatement
    of the Kaolin leaves - against the oracle's own builders AND this package's wisp.ops.spc on the same inputs: dense trees, the
    dilation as the reference's list spells it (23 neighbours: no centre, no -x-y / -x-z / -y-z edge; a corner cell survives through
    clipping), repeated dilation, duplicate points, per-cell attribute means in morton order."""
    from oracle import spc as ospc
    import wisp.ops.spc as pspc
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)          # the reference moves its inputs to the GPU
    leaves = _kaolin_spc_leaves()
    glb = dict(torch=torch, np=np, spc_ops=leaves)
    dilate = glb["dilate_points"] = _reference_function("ops/spc/processing.py", "dilate_points", glb)
    to_octree = _reference_function("ops/spc/conversions.py", "pointcloud_to_octree", glb)
    to_spc = _reference_function("ops/spc/conversions.py", "octree_to_spc", glb)
    dense = _reference_function("ops/spc/constructors.py", "create_dense_octree", glb)
    trilinear = _reference_function("ops/spc/constructors.py", "make_trilinear_spc", glb)

    for level in (1, 2, 3):
        want = ospc.create_dense_octree(level)
        assert np.array_equal(dense(level).numpy(), want)
        assert np.array_equal(pspc.create_dense_octree(level).cpu().numpy(), want)

    for pts, level in ((np.array([[4, 4, 4]]), 3), (np.array([[0, 0, 0], [7, 7, 0]]), 3), (np.array([[1, 2, 3], [1, 2, 4], [9, 9, 9]]), 4)):
        want = ospc.dilate_points(pts, level)
        got_ref = dilate(torch.from_numpy(pts).short(), level).numpy()
        got_pkg = pspc.dilate_points(torch.from_numpy(pts).short(), level).cpu().numpy()
        assert np.array_equal(got_ref, want) and np.array_equal(got_pkg, want)
    shell = {tuple(r) for r in (ospc.dilate_points(np.array([[4, 4, 4]]), 3).astype(int) - 4).tolist()}
    assert len(shell) == 23 and not shell & {(0, 0, 0), (-1, -1, 0), (-1, 0, -1), (0, -1, -1)}

    rng = np.random.default_rng(41)
    cloud = rng.uniform(-1, 1, (400, 3)).astype(np.float32)
    cloud[:50] = cloud[50:100]                                                    # duplicates: several inputs per cell
    cloud[0] = (1.0, -1.0, 1.0)                                                   # the clamp at the upper face
    for level, rounds, width in ((4, 0, 3), (4, 1, 3), (3, 2, 3), (5, 0, 5)):
        att = rng.normal(size=(400, width)).astype(np.float32)
        want = ospc.pointcloud_to_octree(cloud, level, dilate=rounds)
        want_tree, want_att = ospc.pointcloud_to_octree(cloud, level, attributes=att, dilate=0)
        C, A = torch.from_numpy(cloud), torch.from_numpy(att)
        assert np.array_equal(pspc.pointcloud_to_octree(C, level, dilate=rounds).cpu().numpy(), want)
        tree, mean = pspc.pointcloud_to_octree(C, level, attributes=A)
        assert np.array_equal(tree.cpu().numpy(), want_tree)
        np.testing.assert_allclose(mean.cpu().numpy(), want_att, atol=1e-6, rtol=0)
        assert np.array_equal(to_octree(C, level, dilate=rounds).numpy(), want)
        if width == 3:                                                            # the reference's accumulator is zeros_like(unique): F == 3
            tree, mean = to_octree(C, level, attributes=A)
            assert np.array_equal(tree.numpy(), want_tree)
            np.testing.assert_allclose(mean.numpy(), want_att, atol=1e-6, rtol=0)
        assert want_att.shape[0] == ospc.octree_to_spc(want_tree)[1][0, level] < 400

    tree = torch.from_numpy(ospc.pointcloud_to_octree(cloud, 4, dilate=1))
    pts, pyr, ex = ospc.octree_to_spc(tree.numpy())
    for got in (to_spc(tree), pspc.octree_to_spc(tree)):
        assert np.array_equal(got[0].cpu().numpy(), pts) and np.array_equal(got[1].cpu().numpy(), pyr)
        assert np.array_equal(got[2].cpu().numpy(), ex)
    pd, pyd = ospc.make_dual(pts, pyr)
    tr, par = ospc.make_trinkets(pts, pyr, pd, pyd)
    for got in (trilinear(torch.from_numpy(pts), torch.from_numpy(pyr)), pspc.make_trilinear_spc(torch.from_numpy(pts), torch.from_numpy(pyr))):
        for a, b in zip(got, (pd, pyd, tr, par)):
            assert np.array_equal(a.cpu().numpy(), np.asarray(b))


def test_validation_metric_and_log_line_are_what_the_reference_computes_and_parses():
    """(1) psnr (ops/image/metrics.py:19-37), the function body compiled from the reference file (its module imports skimage), equals
    this package's wisp.ops.image.psnr and the oracle's, including the range asserts.  (2) evaluate_psnr's log line goes through the
    reference tests' OWN scraper (tests/test_utils.py:55-92, executed from where it lies) and comes back as the same epoch and value -
    the contract the reference's PSNR-floor tests (tests/apps/test_nerf.py) rest on.  (3) the chunked render concatenates like
    OfflineRenderer.render (tracker/offline_renderer.py:170-191): any chunk size, same image."""
    from oracle import nerf as onerf
    from wisp.core import Rays, RenderBuffer
    from wisp.ops.image import psnr
    from wisp.trainers import evaluate_psnr, render
    ref_psnr = _reference_function("ops/image/metrics.py", "psnr", dict(torch=torch, np=np))
    rng = np.random.default_rng(51)
    a, b = torch.from_numpy(rng.uniform(0, 1, (20, 30, 3)).astype(np.float32)), torch.from_numpy(rng.uniform(0, 1, (20, 30, 3)).astype(np.float32))
    assert ref_psnr(a, b) == psnr(a, b) == onerf.psnr(a, b) and 5.0 < psnr(a, b) < 12.0
    for bad in (a + 0.2, a - 0.2):
        for fn in (ref_psnr, psnr):
            with pytest.raises(AssertionError):
                fn(bad, b)

    def _image(rays):          # exactly rounded ops only: a chunk must give the same bits as the whole (CPU sigmoid is length dependent)
        return (rays.origins * 0.125 + rays.dirs * 0.0625 + 0.5).clamp(0.0, 1.0)

    class Tracer:
        def __call__(self, nef, rays=None, lod_idx=None, channels=None):
            return RenderBuffer(rgb=_image(rays), alpha=rays.dirs[:, :1].abs())

    pipe = types.SimpleNamespace(tracer=Tracer(), nef=types.SimpleNamespace(grid=types.SimpleNamespace(num_lods=16)),
                                 eval=lambda: None, train=lambda: None)
    views = []
    for v in range(3):
        rays = Rays(torch.from_numpy(rng.normal(size=(600, 3)).astype(np.float32)), torch.from_numpy(rng.normal(size=(600, 3)).astype(np.float32)))
        views.append((rays, (_image(rays) + 0.02 * torch.from_numpy(rng.normal(size=(600, 3)).astype(np.float32))).clamp(0, 1)))
    whole = render(pipe, views[0][0], render_batch=0)
    for chunk in (1000, 600, 599, 7):
        part = render(pipe, views[0][0], render_batch=chunk)
        assert torch.equal(part.rgb, whole.rgb) and torch.equal(part.alpha, whole.alpha)
    mean, line = evaluate_psnr(pipe, views, epoch=40, max_epochs=50, render_batch=256)
    assert abs(mean - np.mean([ref_psnr(_image(r), g) for r, g in views])) < 1e-9 and 30.0 < mean < 40.0
    scraper = {"re": __import__("re"), "defaultdict": __import__("collections").defaultdict}
    for fn in ("_get_metric_from_log_line", "collect_metrics_from_log"):
        path = "/root/reference/tests/test_utils.py"
        import ast
        tree = ast.parse(open(path).read(), path)
        node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == fn)
        exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), scraper)
    log = "some other output\n" + line + "\nEPOCH 50/50 | lod15 psnr: 12.34\n"
    got = scraper["collect_metrics_from_log"](log, ["psnr"])
    assert got[40]["psnr"] == "{:.2f}".format(mean) and got[50]["psnr"] == "12.34" and set(got) == {40, 50}


def test_render_buffer_behaves_like_the_reference_class(monkeypatch):
    """wisp.core.RenderBuffer against the reference class (core/render_buffer.py:21-439) executed where it lies, its channel
    definitions (core/channels.py, core/channel_fn.py) registered for the duration of the test: construction with custom channels,
    iteration order, channel queries, rgba get / set, `+` / cat incl. a channel missing on one side, custom channels on one side and
    [N,1] meeting [N], mean (boolean `hit` included), reshape / transpose / scale, dtype and device moves, numpy_dict, exr_dict
    (`rgb` becomes `default`), image(), and the depth-ordered blend with and without alpha - identical channel sets, shapes, dtypes
    and values."""
    import wisp.core                                                               # noqa: F401  (parent package of the registered modules)
    from wisp.core import RenderBuffer as Mine

    def load(name, rel):
        mod = types.ModuleType(name)
        mod.__file__ = os.path.join(REF, rel)
        monkeypatch.setitem(sys.modules, name, mod)
        exec(compile(open(mod.__file__).read(), mod.__file__, "exec"), mod.__dict__)
        return mod

    load("wisp.core.channel_fn", "core/channel_fn.py")
    kit = load("wisp.core.channels", "core/channels.py").channels_starter_kit()
    Ref = _exec_reference("core/render_buffer.py")["RenderBuffer"]

    def make(cls, n=12, extra=True, depth=True, seed=0, alpha=True):
        g = torch.Generator().manual_seed(seed)
        kw = dict(rgb=torch.rand(n, 3, generator=g))
        if alpha:
            kw["alpha"] = torch.rand(n, 1, generator=g)
        if depth:
            kw["depth"] = torch.rand(n, 1, generator=g)
        if extra:
            kw.update(hit=torch.rand(n, 1, generator=g) > 0.5, normal=torch.randn(n, 3, generator=g))
        return cls(**kw)

    def live(rb):
        return {k: v for k, v in iter(rb) if v is not None}

    def check(fn, what):
        torch.manual_seed(77)
        a = live(fn(Ref))
        torch.manual_seed(77)                       # some cases draw fresh channels: the same draws for both classes
        b = live(fn(Mine))
        assert set(a) == set(b), (what, sorted(a), sorted(b))
        for k in a:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), (what, k)

    a, b = make(Ref), make(Mine)
    assert [k for k, _ in iter(a)] == [k for k, _ in iter(b)] and a.channels == b.channels
    assert a.has_channel("hit") and b.has_channel("hit") and not a.has_channel("zz") and not b.has_channel("zz")
    assert a.get_channel("zz") is None and b.get_channel("zz") is None and b.zz is None
    assert torch.equal(a.rgba, b.rgba) and make(Ref, alpha=False).rgba is None and make(Mine, alpha=False).rgba is None

    def set_rgba(cls):
        rb = make(cls)
        rb.rgba = torch.full((12, 4), 0.25)
        return rb

    flat_hit = lambda cls: cls(rgb=torch.ones(5, 3), hit=torch.ones(5, dtype=torch.bool))
    cases = {
        "construct": make,
        "rgba setter": set_rgba,
        "add": lambda c: make(c) + make(c, seed=1),
        "add to empty": lambda c: c() + make(c),
        "add empty": lambda c: make(c) + c(),
        "add, depth missing on one side": lambda c: make(c, depth=False) + make(c, seed=1),
        "add, custom channels on one side": lambda c: make(c, extra=False) + make(c, seed=1),
        "cat [N,1] with [N]": lambda c: make(c, 5).cat(flat_hit(c)),
        "cat [N] with [N,1]": lambda c: flat_hit(c).cat(make(c, 5)),
        "cat dim 1": lambda c: make(c).reshape(3, 4, -1).cat(make(c, seed=1).reshape(3, 4, -1), dim=1),
        "mean": lambda c: c.mean(make(c), make(c, seed=1), make(c, seed=2, depth=False)),
        "reshape": lambda c: make(c).reshape(3, 4, -1),
        "transpose": lambda c: make(c).reshape(3, 4, -1).transpose(),
        "scale": lambda c: make(c, extra=False).reshape(3, 4, -1).scale((6, 8)),
        "image": lambda c: make(c).reshape(3, 4, -1).image(),
        "image without hit / normal": lambda c: make(c, extra=False).image(),
        "byte": lambda c: make(c).byte(), "half": lambda c: make(c).half(), "float": lambda c: make(c).half().float(),
        "double": lambda c: make(c).double(), "detach": lambda c: make(c).detach(), "cpu": lambda c: make(c).cpu(),
        "to": lambda c: make(c).to(torch.float64),
        "blend, starter kit": lambda c: make(c, extra=False).blend(make(c, extra=False, seed=1), channel_kit=kit),
        "blend, default rule": lambda c: make(c, extra=False).blend(make(c, extra=False, seed=1), channel_kit={}),
        "blend without alpha": lambda c: make(c, extra=False, alpha=False).blend(make(c, extra=False, seed=1, alpha=False), channel_kit=kit),
        "blend, channel on one side": lambda c: make(c, extra=False).blend(c(depth=torch.rand(12, 1), alpha=torch.rand(12, 1), err=torch.rand(12, 2)),
                                                                           channel_kit=kit),
    }
    for what, fn in cases.items():
        check(fn, what)
    for cls in (Ref, Mine):
        with pytest.raises(AssertionError):
            make(cls, depth=False).blend(make(cls), channel_kit=kit)
    na, nb = make(Ref).numpy_dict(), make(Mine).numpy_dict()
    ea, eb = make(Ref).exr_dict(), make(Mine).exr_dict()
    assert sorted(na) == sorted(nb) and sorted(ea) == sorted(eb) and "default" in eb and "rgb" not in eb
    assert all(np.array_equal(na[k], nb[k]) for k in na) and all(np.array_equal(ea[k], eb[k]) for k in ea)


def test_neural_field_dispatch_behaves_like_the_reference_base_class():
    """BaseNeuralField (models/nefs/base_nef.py:16-202) executed where it lies against wisp.models.nefs.BaseNeuralField, the same toy
    field subclassed from each: which functions run for which requested channels (the one covering most channels first, each at most
    once), required / optional argument forwarding by signature, the return type by the TYPE of `channels` (str -> tensor, list -> list
    in order, set / None -> dict), get_forward_function, and the four error cases with their messages."""
    from wisp.models.nefs import BaseNeuralField as Mine
    Ref = _exec_reference("models/nefs/base_nef.py")["BaseNeuralField"]
    calls = []

    def field(base):
        class Toy(base):
            def register_forward_functions(self):
                self._register_forward_function(self.rgba, ["density", "rgb"])
                self._register_forward_function(self.sdf, "sdf")
                self._register_forward_function(self.both, ["rgb", "normal", "extra"])

            def rgba(self, coords, ray_d, lod_idx=None):
                calls.append(("rgba", lod_idx))
                return dict(rgb=coords + ray_d, density=coords.sum(-1, keepdim=True) * (1 if lod_idx is None else lod_idx))

            def sdf(self, coords, scale=2.0):
                calls.append(("sdf", scale))
                return dict(sdf=coords.norm(dim=-1, keepdim=True) * scale)

            def both(self, coords):
                calls.append(("both", None))
                return dict(rgb=-coords, normal=coords * 2, extra=coords[:, :1])
        return Toy()

    x, d = torch.arange(12.0).reshape(4, 3), torch.ones(4, 3)

    def run(nef, **kw):
        calls.clear()
        try:
            out = nef(**kw)
        except Exception as e:                                   # noqa: BLE001 - the reference raises bare Exception
            return ("raised", type(e).__name__, str(e)), list(calls)
        return out, list(calls)

    def same(a, b):
        if isinstance(a, torch.Tensor):
            return isinstance(b, torch.Tensor) and torch.equal(a, b)
        if isinstance(a, dict):
            return isinstance(b, dict) and set(a) == set(b) and all(same(a[k], b[k]) for k in a)
        if isinstance(a, list):
            return isinstance(b, list) and len(a) == len(b) and all(same(p, q) for p, q in zip(a, b))
        return type(a) is type(b) and a == b

    ref, mine = field(Ref), field(Mine)
    assert ref.get_supported_channels() == mine.get_supported_channels() == {"density", "rgb", "sdf", "normal", "extra"}
    requests = [
        dict(channels="rgb", coords=x, ray_d=d), dict(channels="density", coords=x, ray_d=d, lod_idx=3),
        dict(channels=["rgb", "density"], coords=x, ray_d=d), dict(channels=["density", "rgb"], coords=x, ray_d=d),
        dict(channels={"rgb", "density"}, coords=x, ray_d=d), dict(channels=None, coords=x, ray_d=d, scale=0.5),
        dict(channels="sdf", coords=x), dict(channels="sdf", coords=x, scale=3.0, unused=1),
        dict(channels=["normal", "rgb", "extra"], coords=x, ray_d=d),          # `both` covers three: runs first, rgba not needed
        dict(channels=["normal", "density"], coords=x, ray_d=d), dict(channels={"sdf", "normal"}, coords=x),
        dict(channels=["rgb"], coords=x),                                         # rgba lacks ray_d ...
        dict(channels="density", coords=x),                                      # ... required argument missing
        dict(channels="colour", coords=x, ray_d=d),                              # unsupported channel
        dict(channels=("rgb",), coords=x, ray_d=d),                              # tuple: invalid type
        dict(channels=[], coords=x),
    ]
    for kw in requests:
        (ra, ca), (rb, cb) = run(ref, **kw), run(mine, **kw)
        assert ca == cb, (kw["channels"], ca, cb)
        if isinstance(ra, tuple) and ra and ra[0] == "raised":
            assert isinstance(rb, tuple) and rb[0] == "raised" and rb[1] == ra[1], (kw["channels"], ra, rb)
            assert ra[2].replace("Toy", "") == rb[2].replace("Toy", ""), (ra[2], rb[2])
        else:
            assert same(ra, rb), (kw["channels"], ra, rb)
    for ch in ("rgb", "sdf", "normal"):
        fa, fb = ref.get_forward_function(ch), mine.get_forward_function(ch)
        kw = dict(coords=x) if ch != "rgb" else dict(coords=x, ray_d=d)
        assert torch.equal(fa(**kw), fb(**kw))
    for nef in (ref, mine):
        with pytest.raises(Exception, match="not supported"):
            nef.get_forward_function("colour")


def test_batch_types_and_ray_sampler_behave_like_the_reference(monkeypatch):
    """Batch / MultiviewBatch / SDFBatch (datasets/batch.py:19-110) executed where they lie (attrdict.AttrDict and kaolin's Camera, third
    party, stubbed: a dict whose items also read as attributes) and SampleRays.__call__ (datasets/transforms/ray_sampler.py:24-35), the
    method body compiled from the reference file - against wisp.datasets: field lists and order, attribute access, ray_values /
    coord_values, and - with the same seed of the global generator - the very same sampled rays and colours."""
    from wisp.core import Rays
    from wisp.datasets import Batch as MyBatch, MultiviewBatch as MyMV, SDFBatch as MySDF, SampleRays as MySampler

    class AttrDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    monkeypatch.setitem(sys.modules, "attrdict", types.SimpleNamespace(AttrDict=AttrDict))
    cam_mod = types.SimpleNamespace(Camera=object)
    for name in ("kaolin", "kaolin.render", "kaolin.render.camera"):
        monkeypatch.setitem(sys.modules, name, types.SimpleNamespace(render=types.SimpleNamespace(camera=cam_mod), camera=cam_mod, Camera=object))
    ref = _exec_reference("datasets/batch.py")
    g = torch.Generator().manual_seed(2)
    rays = Rays(torch.rand(500, 3, generator=g), torch.rand(500, 3, generator=g), dist_min=0.5, dist_max=6.0)
    rgb = torch.rand(500, 3, generator=g)
    for cams in (None, ["cam0"]):
        a, b = ref["MultiviewBatch"](rays=rays, cameras=cams, rgb=rgb), MyMV(rays=rays, cameras=cams, rgb=rgb)
        assert a.fields == b.fields == ["rays", "cameras", "rgb"] and list(a.ray_values()) == list(b.ray_values()) == ["rgb"]
        assert b.rays is rays and b["rgb"] is b.rgb is a.rgb and b.cameras == a.cameras
    assert ref["MultiviewBatch"](rays=rays).ray_values() == MyMV(rays=rays).ray_values() == {}
    b = MyMV(rays=rays, rgb=rgb)
    b.depth = rgb[:, :1]                                                           # attribute writes are item writes
    assert b["depth"] is rgb[:, :1] or torch.equal(b["depth"], rgb[:, :1])
    with pytest.raises(AttributeError):
        b.nothing_here
    co, sd, nrm = torch.rand(40, 3, generator=g), torch.rand(40, 1, generator=g), torch.rand(40, 3, generator=g)
    for kw in (dict(), dict(rgb=co * 0.5), dict(normals=nrm), dict(rgb=co * 0.5, normals=nrm)):
        a, b = ref["SDFBatch"](coords=co, sdf=sd, **kw), MySDF(coords=co, sdf=sd, **kw)
        assert a.fields == b.fields == ["coords", "sdf", "rgb", "normals"] and list(a.coord_values()) == list(b.coord_values())
        assert all(a.coord_values()[k] is b.coord_values()[k] for k in a.coord_values()) and b.coords is co and b.sdf is sd
    assert ref["Batch"](x=1, y=2).fields == MyBatch(x=1, y=2).fields == ["x", "y"] and MyBatch({"q": 3}).q == 3

    call = _reference_method("datasets/transforms/ray_sampler.py", "SampleRays", "__call__",
                             dict(torch=torch, MultiviewBatch=ref["MultiviewBatch"]))
    for n in (64, 500, 1200):                                                       # fewer, as many, more than the view has (with repeats)
        torch.manual_seed(123)
        want = call(types.SimpleNamespace(num_samples=n), ref["MultiviewBatch"](rays=rays, rgb=rgb))
        torch.manual_seed(123)
        got = MySampler(n)(MyMV(rays=rays, rgb=rgb))
        assert list(want) == list(got) == ["rays", "rgb"] and got["rgb"].shape == (n, 3) and got["rgb"].is_contiguous()
        assert torch.equal(got["rgb"], want["rgb"]) and torch.equal(got["rays"].origins, want["rays"].origins)
        assert torch.equal(got["rays"].dirs, want["rays"].dirs)
        assert (got["rays"].dist_min, got["rays"].dist_max) == (want["rays"].dist_min, want["rays"].dist_max) == (0.5, 6.0)
    s = MySampler(8)
    s.set_num_samples(32)
    assert s.num_samples == 32


def test_mesh_helpers_equal_the_reference_function_bodies():
    """per_face_normals, area_weighted_distribution, random_face, sample_surface, sample_near_surface, sample_uniform, point_sample,
    normalize (all four modes) and barycentric_coordinates (wisp/ops/mesh/*.py), each function compiled from its reference file -
    against wisp.ops.mesh on an octahedron with unequal faces, same seed of the global generator: identical normals, face
    probabilities, drawn faces, surface / near-surface / uniform samples and their order in point_sample, normalised vertices,
    barycentric coordinates (inside, outside and on an edge)."""
    from wisp.ops import mesh as mine
    glb = dict(torch=torch)
    for name in ("per_face_normals", "area_weighted_distribution", "random_face", "sample_surface", "sample_near_surface",
                 "sample_uniform", "point_sample", "normalize", "barycentric_coordinates"):
        glb[name] = _reference_function(f"ops/mesh/{name}.py", name, glb)
    V = torch.tensor([[1.3, 0.1, 0.0], [-0.7, 0.0, 0.2], [0.0, 2.1, 0.1], [0.1, -0.9, 0.0], [0.0, 0.2, 0.8], [0.2, 0.0, -1.7]])
    F = torch.tensor([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])

    def both(fn, *args, **kw):
        torch.manual_seed(5)
        a = glb[fn](*args, **kw)
        torch.manual_seed(5)
        return a, getattr(mine, fn)(*args, **kw)

    a, b = both("per_face_normals", V, F)
    assert torch.equal(a, b) and float(a.norm(dim=1).min()) > 0.5
    a, b = both("area_weighted_distribution", V, F)
    assert torch.equal(a.probs, b.probs) and float(a.probs.max() / a.probs.min()) > 2.0
    for fn, args in (("random_face", (V, F, 300)), ("sample_surface", (V, F, 300))):
        a, b = both(fn, *args)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    a, b = both("sample_near_surface", V, F, 300, variance=0.05)
    assert torch.equal(a, b)
    a, b = both("sample_uniform", 200)
    assert torch.equal(a, b) and a.shape == (200, 3) and float(a.abs().max()) <= 1.0
    for techniques in (["trace"], ["rand"], ["near", "trace", "rand"], ["rand", "unknown", "near"]):
        a, b = both("point_sample", V, F, techniques, 100)
        assert torch.equal(a, b) and a.shape[0] == 100 * sum(t in ("trace", "near", "rand") for t in techniques)
    for mode in ("sphere", "aabb", "planar", "none"):
        a, b = both("normalize", V.clone(), F, mode)
        assert torch.equal(a[0], b[0]) and a[1] is F and b[1] is F, mode
    assert abs(float(glb["normalize"](V.clone(), F, "sphere")[0].norm(dim=1).max()) - 1.0) < 1e-6
    tri = V[F[torch.tensor([0, 3, 5, 6])]]
    w = torch.tensor([[0.2, 0.3, 0.5], [1.4, -0.2, -0.2], [0.5, 0.5, 0.0], [-0.3, 0.9, 0.4]])
    pts = (tri * w[:, :, None]).sum(1)
    a, b = both("barycentric_coordinates", pts, tri[:, 0], tri[:, 1], tri[:, 2])
    assert torch.equal(a, b) and torch.allclose(a[0], w[0], atol=1e-5) and float(a.min()) >= 0.0 and float(a.max()) <= 1.0


@pytest.mark.parametrize("multiscale", ["cat", "sum"])
def test_oracle_triplanar_grid_equals_the_reference_methods(multiscale):
    """TriplanarFeatureVolume.forward and TriplanarGrid.interpolate / _interpolate (models/grids/triplanar_grid.py:98-143,205-233),
    the method bodies compiled from the reference file and run on the CPU (the reference's arithmetic here IS torch's grid_sample) -
    against oracle.triplanar.interpolate, what the one-launch HIP triplanar kernel is compared with: which coordinate pair addresses
    which plane, reflection padding outside [-1, 1], the plane-major [x | y | z] feature layout after stack / permute / reshape,
    'cat' / 'sum' over levels, [B,3] and [B,S,3] inputs."""
    import torch.nn.functional as F_
    from oracle import triplanar as otri
    glb = dict(torch=torch, F=F_)
    forward = _reference_method("models/grids/triplanar_grid.py", "TriplanarFeatureVolume", "forward", glb)
    interp_one = _reference_method("models/grids/triplanar_grid.py", "TriplanarGrid", "_interpolate", glb)
    interp = _reference_method("models/grids/triplanar_grid.py", "TriplanarGrid", "interpolate", glb)
    torch.manual_seed(61)
    fdim, sizes = 4, (8, 16, 32)
    volumes = [tuple(torch.randn(1, fdim, s + 1, s + 1) for _ in range(3)) for s in sizes]

    class Volume:
        def __init__(self, planes):
            self.fmx, self.fmy, self.fmz = planes
            self.fdim, self.padding_mode = fdim, 'reflection'

        def __call__(self, x):
            return forward(self, x)

    me = types.SimpleNamespace(features=[Volume(v) for v in volumes], multiscale_type=multiscale, interpolation_type='linear')
    me._interpolate = lambda coords, feats, lod_idx: interp_one(me, coords, feats, lod_idx)
    rng = np.random.default_rng(62)
    x = torch.from_numpy(rng.uniform(-1.2, 1.2, (240, 3)).astype(np.float32))           # some beyond the planes: reflection
    for lod_idx in (0, 2):
        width = 3 * fdim * (1 if multiscale == 'sum' else lod_idx + 1)
        want = otri.interpolate(volumes, x, lod_idx, multiscale)
        for coords in (x, x.reshape(48, 5, 3)):
            got = interp(me, coords, lod_idx)
            # [B,3] is inflated to [B,1,3]; only the 'sum' branch restores the caller's shape (a reference quirk this package keeps)
            assert got.shape == ((*coords.shape[:-1], width) if (multiscale == 'sum' or coords.ndim == 3) else (coords.shape[0], 1, width))
            assert torch.allclose(got.reshape(-1, width), want, atol=1e-6, rtol=0), float((got.reshape(-1, width) - want).abs().max())
    one = forward(me.features[0], x[:7, None, :])                                          # [N, 1, 3, fdim]: plane axis before features
    ref_planes = otri.volume_forward(*volumes[0], x[:7])
    assert one.shape == (7, 1, 3, fdim) and torch.allclose(one[:, 0], ref_planes, atol=1e-6)
    assert not torch.allclose(ref_planes[:, 0], ref_planes[:, 1], atol=1e-3)              # the three planes really differ


def test_octree_as_construction_and_bookkeeping_equal_the_reference_class(monkeypatch):
    """The reference's OWN OctreeAS (accelstructs/octree_as.py:37-144,431-441) executed where it lies - over this package's
    wisp.ops.spc / base_as and the oracle's restatement of kaolin's unbatched_points_to_octree - against wisp.accelstructs.OctreeAS:
    __init__, make_dense, from_quantized_points (unsorted, with duplicates), from_pointcloud, and AxisAlignedBBoxAS; octree bytes,
    point hierarchy, pyramid, prefix sums, max_level, occupancy(), capacity(), name(), the empty `extent`."""
    from oracle import spc as ospc
    import wisp.accelstructs as mine
    stubs = _kaolin_stub()
    stubs["kaolin.ops.spc"].unbatched_points_to_octree = lambda p, level, sorted=False: torch.from_numpy(
        ospc.points_to_octree(p.cpu().numpy(), level))
    stubs["kaolin"].render = types.ModuleType("kaolin.render")
    stubs["kaolin.render"] = stubs["kaolin"].render
    stubs["kaolin.render.spc"] = stubs["kaolin"].render.spc = types.ModuleType("kaolin.render.spc")
    for name, mod in stubs.items():
        monkeypatch.setitem(sys.modules, name, mod)
    ref = _exec_reference("accelstructs/octree_as.py")
    Ref, Mine = ref["OctreeAS"], mine.OctreeAS
    rng = np.random.default_rng(71)
    q = torch.from_numpy(rng.integers(0, 32, size=(400, 3)).astype(np.int16))
    q[:40] = q[40:80]
    cloud = torch.from_numpy(rng.uniform(-1, 1, (500, 3)).astype(np.float32))
    builds = {
        "make_dense(2)": lambda c: c.make_dense(2),
        "from_quantized_points": lambda c: c.from_quantized_points(q, 5),
        "from_pointcloud": lambda c: c.from_pointcloud(cloud, 4),
        "__init__": lambda c: c(torch.from_numpy(ospc.points_to_octree(q.numpy(), 5))),
    }
    for what, build in builds.items():
        a, b = build(Ref), build(Mine)
        assert isinstance(a, Ref) and isinstance(b, Mine), what
        for field in ("octree", "points", "pyramid", "prefix"):
            x, y = getattr(a, field).cpu(), getattr(b, field).cpu()
            assert x.shape == y.shape and torch.equal(x.long(), y.long()), (what, field)
        assert a.max_level == b.max_level and a.occupancy() == b.occupancy() and a.capacity() == b.capacity(), what
        assert a.name() == b.name() == "Octree" and a.extent == b.extent == {}
    dense = builds["make_dense(2)"](Mine)
    assert dense.occupancy() == [1, 8] and dense.capacity() == [1, 8] and dense.max_level == 2
    exec(compile(open(os.path.join(REF, "accelstructs/aabb_as.py")).read(), "aabb_as.py", "exec"),
         box := {"__name__": "reference_aabb"})                        # its `from wisp.accelstructs.octree_as import OctreeAS` = this package's
    a, b = box["AxisAlignedBBoxAS"](), mine.AxisAlignedBBoxAS()
    assert a.name() == b.name() == "AABB" and torch.equal(a.octree.cpu(), b.octree.cpu()) and a.max_level == b.max_level == 1


def _reference_nerf_stack(monkeypatch, half_autocast=False):
    """The reference's own OctreeAS / HashGrid / NeuralRadianceField / PackedRFTracer modules executed where they lie, over the oracle's
    Kaolin leaves and the reference's hash-grid kernel bodies built for the host (forward and backward).
    half_autocast: the stack is going to run under torch.autocast('cpu', dtype=torch.float16) standing in for the reference's
    torch.cuda.amp.autocast() (base_trainer.py:338).  Two things of ops/grid.py look at the CUDA autocast state and the kernel's
    dtype dispatch and need a CPU counterpart: `torch.is_autocast_enabled()` (grid.py:87) answers for the CPU context, and the
    kernel entry points take the fp16 table `codebook.half()` hands them - the scalar_t = half instantiation reads half entries,
    blends in float and rounds the result to half (hashgrid_interpolate_cuda.cu:70-79: static_cast<float>(codebook[..]) ...
    static_cast<scalar_t>(feat)), which is the float kernel body on float(half table) with the output rounded once; the backward
    body's float arithmetic on the float-of-half gradient, returned in the table's dtype like the reference's at::zeros_like.
    -> (octree_as module namespace, HashGrid, NeuralRadianceField, PackedRFTracer)"""
    from oracle import nerf as onerf, render as orender, spc as ospc, ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built")
    import wisp
    import wisp.ops
    from wisp.core import Rays
    t = torch.from_numpy

    # ---- Kaolin leaves = the oracle's restatements
    stubs = _kaolin_stub()
    kspc = stubs["kaolin.ops.spc"]
    kspc.unbatched_points_to_octree = lambda p, level, sorted=False: t(ospc.points_to_octree(p.cpu().numpy(), level))
    kspc.unbatched_query = lambda octree, prefix, coords, level, with_parents=False: t(
        ospc.query(octree.numpy(), prefix.numpy(), coords.detach().numpy(), level, with_parents=with_parents))
    krender = types.ModuleType("kaolin.render")
    krs = types.ModuleType("kaolin.render.spc")
    krs.mark_pack_boundaries = krs.mark_first_hit = lambda ridx: t(ospc.mark_pack_boundaries(ridx.numpy()))

    def raytrace(octree, points, pyramid, prefix, origins, dirs, level, return_depth=True, with_exit=False):
        ridx, pidx, depth = ospc.raytrace(octree.numpy(), points.numpy(), pyramid.numpy(), prefix.numpy(), origins.numpy(), dirs.numpy(),
                                          level, with_exit=with_exit)
        return t(ridx), t(pidx), t(depth.copy())
    krs.unbatched_raytrace = raytrace
    krs.exponential_integration = orender.exponential_integration
    krs.sum_reduce = orender.sum_reduce
    krender.spc = stubs["kaolin"].render = krs
    stubs["kaolin"].render = krender
    stubs.update({"kaolin.render": krender, "kaolin.render.spc": krs})

    # ---- wisp._C = the reference's hash-grid kernel bodies built for the host
    def fwd(coords, codebook, first_idx, resolution, bitwidth):
        res = [int(r) for r in resolution.reshape(-1).tolist()]
        assert codebook.dtype == (torch.float16 if half_autocast else torch.float32)
        out = t(ref_lib.hashgrid_forward(coords.numpy(), codebook.detach().float().numpy(), first_idx.numpy(), res, int(bitwidth)))
        return out.to(codebook.dtype)
    native = types.ModuleType("wisp._C")

    def bwd(coords, grad_output, codebook, first_idx, resolution, bitwidth, feature_dim, require_grad_coords):
        res = [int(r) for r in resolution.reshape(-1).tolist()]
        assert grad_output.dtype == codebook.dtype == (torch.float16 if half_autocast else torch.float32)
        g = ref_lib.hashgrid_backward(coords.numpy(), grad_output.contiguous().float().numpy(), codebook.detach().float().numpy(),
                                      first_idx.numpy(), res, int(bitwidth))
        return [torch.empty(0), t(g).to(codebook.dtype)]
    native.ops = types.SimpleNamespace(hashgrid_interpolate_cuda=fwd, hashgrid_interpolate_backward_cuda=bwd)
    stubs["wisp._C"] = native
    for name, mod in stubs.items():
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(wisp, "_C", native, raising=False)
    grid_mod = types.ModuleType("wisp.ops.grid")
    grid_ns = _exec_reference("ops/grid.py")                                       # `import wisp._C as wisp_C` -> the host build
    if half_autocast:
        class _TorchCpuAutocast(_TorchWithoutNvtx):
            def is_autocast_enabled(self, *a):
                return torch.is_autocast_enabled('cpu')
        grid_ns["torch"] = _TorchCpuAutocast()                                     # (the globals the reference's functions look names up in)
    grid_mod.__dict__.update(grid_ns)
    monkeypatch.setitem(sys.modules, "wisp.ops.grid", grid_mod)
    monkeypatch.setattr(wisp.ops, "grid", grid_mod, raising=False)                 # `import wisp.ops.grid as grid_ops` walks attributes

    blas_mod = _exec_reference("accelstructs/octree_as.py")
    RefGrid = _exec_reference("models/grids/hash_grid.py")["HashGrid"]
    RefField = _exec_reference("models/nefs/nerf.py")["NeuralRadianceField"]
    RefTracer = _exec_reference("tracers/packed_rf_tracer.py")["PackedRFTracer"]

    return blas_mod, RefGrid, RefField, RefTracer


@pytest.mark.parametrize("mode,steps", [("ray", 96), ("voxel", 6)])
def test_oracle_render_equals_the_whole_reference_stack_on_the_host(monkeypatch, mode, steps):
    """End to end on the CPU with the reference's OWN classes wired together as an application would: OctreeAS
    (accelstructs/octree_as.py) -> HashGrid.from_geometric (models/grids/hash_grid.py) over the reference's ops/grid.py and its hash-grid
    kernel bodies built for the host (oracle/_ref) -> NeuralRadianceField (models/nefs/nerf.py) -> PackedRFTracer
    (tracers/packed_rf_tracer.py), every module executed where it lies; supplied from outside: the Kaolin leaves (the oracle's
    restatements) and the jitter draw.  Against oracle.nerf.trace with the same parameters - the tracer every GPU end-to-end test is
    compared with.  The pieces are pinned one by one elsewhere; this pins the glue: BaseTracer's argument plumbing, the field's channel
    dispatch, the level the grid marches at, the default lod_idx, table layout and state-dict names."""
    from oracle import nerf as onerf
    from wisp.core import Rays
    t = torch.from_numpy
    blas_mod, RefGrid, RefField, RefTracer = _reference_nerf_stack(monkeypatch)

    rng = np.random.default_rng(81)
    pts = rng.integers(0, 16, size=(500, 3))
    R, bg = 140, (0.2, 0.5, 0.7)
    blas = blas_mod["OctreeAS"].from_quantized_points(t(pts.astype(np.int16)), 4)
    torch.manual_seed(82)
    grid = RefGrid.from_geometric(blas, feature_dim=2, num_lods=4, multiscale_type='cat', feature_std=0.3, codebook_bitwidth=10,
                                  min_grid_res=8, max_grid_res=64)
    nef = RefField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1, bias=True)
    tracer = RefTracer(raymarch_type=mode, num_steps=steps, bg_color=bg)
    o = rng.normal(size=(R, 3)).astype(np.float32)
    o = 3.0 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = -o + rng.normal(size=o.shape).astype(np.float32) * 0.4
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    oblas = onerf.OracleBLAS.from_quantized_points(pts, 4)
    if mode == "ray":
        jit = rng.uniform(size=(R, steps)).astype(np.float32)
        blas_mod["torch"] = _TorchWithDraws(t(jit))                                 # OctreeAS._raymarch_ray's torch.rand(R, N)
    else:                                                                           # 'voxel': N jittered samples in every intersected cell
        from oracle import spc as ospc
        import wisp.ops.spc as package_spc
        nuggets = ospc.raytrace(oblas.octree, oblas.points, oblas.pyramid, oblas.exsum, o, d, 4, True)[0].shape[0]
        jit = rng.uniform(size=(nuggets, steps)).astype(np.float32)
        sampling = _exec_reference("ops/spc/sampling.py")                           # the reference's own helpers, their rand_like injected
        sampling["torch"] = _TorchWithDraws(t(jit))

        class SpcOps:
            sample_from_depth_intervals = staticmethod(sampling["sample_from_depth_intervals"])
            expand_pack_boundary = staticmethod(sampling["expand_pack_boundary"])

            def __getattr__(self, name):
                return getattr(package_spc, name)
        blas_mod["wisp_spc_ops"] = SpcOps()
    with torch.no_grad():
        rb = tracer(nef, rays=Rays(t(o), t(d), dist_min=1.0, dist_max=5.0), channels={"rgb", "depth", "alpha", "hit"})

    # ---- the oracle with the very same parameters
    res = [int(r) for r in grid.resolutions]
    onef = onerf.OracleNeRF(res, 2, 10, 'cat', 0.3, 64, 1, True, 4)
    missing = onef.load_state_dict(nef.state_dict(), strict=False)
    # every learnable tensor has its counterpart under the same name; what the oracle does not keep are the reference's buffers
    assert not missing.missing_keys and set(missing.unexpected_keys) == {"view_embedder.bands", "grid.codebook.begin_idxes",
                                                                         "grid.codebook.num_feats"}, missing
    assert torch.equal(nef.state_dict()["grid.codebook.begin_idxes"].long(), torch.as_tensor(onef.begin_idxes).long())
    assert np.array_equal(oblas.octree, blas.octree.numpy()) and len(res) == 4 and res[0] == 8 and res[-1] == 64
    with torch.no_grad():
        want = onerf.trace(onef, oblas, t(o), t(d), 1.0, 5.0, steps, jit, bg, mode, with_depth=True)
    assert tracer.prev_num_samples == want["raymarch"]["ridx"].shape[0] > 500
    assert torch.equal(rb.hit, want["hit"]) and 20 < int(rb.hit.sum()) < R
    # sample depths differ in the last bit (the oracle restates them as the GPU evaluates them, torch's CPU kernels order a few
    # operations differently: 1e-6, see the raymarch pin above); depth = sum of weight x sample depth carries that relatively
    for name, rtol in (("rgb", 0.0), ("alpha", 0.0), ("depth", 2e-6)):
        got, ref = getattr(rb, name), want[name]
        assert got.shape == ref.shape and torch.allclose(got, ref, atol=2e-6, rtol=rtol), (name, float((got - ref).abs().max()))
    assert float(rb.rgb.std()) > 0.05 and float(rb.depth.max()) > 1.0


def test_oracle_training_steps_equal_the_whole_reference_stack_on_the_host(monkeypatch):
    """Three optimisation steps through the reference's own classes on the CPU: PackedRFTracer -> NeuralRadianceField -> HashGrid, the
    gradient going back through the reference's HashGridInterpolate autograd function (ops/grid.py:77-126) into its backward kernel body
    built for the host, its decoders and the compositing; huber loss over the rays; AdamW with the reference's parameter groups - next
    to oracle.nerf.train_step from the same initial state, batches and jitter: the same loss at every step, the same gradient for every
    parameter at the first step, the same parameters after the third."""
    from oracle import nerf as onerf
    from wisp.core import Rays
    t = torch.from_numpy
    blas_mod, RefGrid, RefField, RefTracer = _reference_nerf_stack(monkeypatch)
    rng = np.random.default_rng(91)
    pts = rng.integers(0, 16, size=(600, 3))
    steps, R, bg = 64, 96, (0.0, 0.0, 0.0)
    blas = blas_mod["OctreeAS"].from_quantized_points(t(pts.astype(np.int16)), 4)
    torch.manual_seed(92)
    grid = RefGrid.from_geometric(blas, feature_dim=2, num_lods=4, multiscale_type='cat', feature_std=0.3, codebook_bitwidth=10,
                                  min_grid_res=8, max_grid_res=64)
    nef = RefField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1, bias=True)
    tracer = RefTracer(raymarch_type='ray', num_steps=steps, bg_color=bg)
    onef = onerf.OracleNeRF([int(r) for r in grid.resolutions], 2, 10, 'cat', 0.3, 64, 1, True, 4)
    onef.load_state_dict(nef.state_dict(), strict=False)
    oblas = onerf.OracleBLAS.from_quantized_points(pts, 4)
    opt_ref = onerf.make_optimizer(nef, lr=1e-2, grid_lr_weight=10.0)             # groups by the 'decoder' / 'grid' names: same rule for both
    opt_ora = onerf.make_optimizer(onef, lr=1e-2, grid_lr_weight=10.0)
    names = [n for n, p in nef.named_parameters() if p.requires_grad]
    assert names == [n for n, p in onef.named_parameters() if p.requires_grad] and "grid.codebook.feats" in names
    before = {n: p.detach().clone() for n, p in nef.named_parameters()}
    for step in range(3):
        o = rng.normal(size=(R, 3)).astype(np.float32)
        o = 3.0 * o / np.linalg.norm(o, axis=1, keepdims=True)
        d = -o + rng.normal(size=o.shape).astype(np.float32) * 0.4
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        gts = t(rng.uniform(0, 1, (R, 3)).astype(np.float32))
        jit = rng.uniform(size=(R, steps)).astype(np.float32)
        blas_mod["torch"] = _TorchWithDraws(t(jit))
        opt_ref.zero_grad()
        rb = tracer(nef, rays=Rays(t(o), t(d), dist_min=1.0, dist_max=5.0), channels=["rgb"])
        loss = torch.nn.functional.smooth_l1_loss(rb.rgb, gts, reduction='none').mean()   # multiview_trainer.py:149-154 (pinned elsewhere)
        loss.backward()
        if step == 0:
            grads_ref = {n: p.grad.detach().clone() for n, p in nef.named_parameters() if p.grad is not None}
        opt_ref.step()
        if step == 0:                                                              # the oracle's gradient, before its own step changes anything
            probe = onerf.OracleNeRF([int(r) for r in grid.resolutions], 2, 10, 'cat', 0.3, 64, 1, True, 4)
            probe.load_state_dict({k: v for k, v in before.items()}, strict=False)
            res = onerf.trace(probe, oblas, t(o), t(d), 1.0, 5.0, steps, jit, bg, 'ray', with_depth=False)
            torch.nn.functional.smooth_l1_loss(res["rgb"], gts, reduction='none').mean().backward()
            for n, p in probe.named_parameters():
                if p.requires_grad:
                    assert n in grads_ref and float(grads_ref[n].abs().max()) > 0, n
                    scale = float(grads_ref[n].abs().max())
                    assert torch.allclose(p.grad, grads_ref[n], atol=3e-5 * scale + 1e-9, rtol=0), (n, float((p.grad - grads_ref[n]).abs().max()), scale)
        want_loss, want_samples = onerf.train_step(onef, oblas, opt_ora, t(o), t(d), gts, 1.0, 5.0, steps, jit, bg, 'ray', 'huber')
        assert tracer.prev_num_samples == want_samples > 300
        assert abs(float(loss) - want_loss) < 2e-6, (step, float(loss), want_loss)
    moved = 0.0
    theirs = dict(onef.named_parameters())
    for n, p in nef.named_parameters():
        if not p.requires_grad:                                                    # the reference's frozen `view_embedder.bands`
            continue
        diff, travelled = (p - theirs[n]).abs(), (p - before[n]).abs()
        # Adam divides by sqrt(v): where a gradient is at add-order-noise level the normalised step amplifies that noise, so a few
        # entries of the table differ by more than the bulk - never by more than a fraction of a percent of their own movement
        assert float((diff > 2e-5).float().mean()) < 1e-3 and float(diff.max()) < 1e-4, (n, float(diff.max()))
        assert float((diff / travelled.clamp_min(1e-9)).max()) < 5e-3 or float(diff.max()) < 2e-6, n
        moved = max(moved, float(travelled.max()))
    assert moved > 5e-3                                                            # three real AdamW steps at lr 1e-2


def test_half_rounding_oracle_equals_the_reference_stack_under_fp16_autocast_on_the_host(monkeypatch):
    """oracle.nerf's fp16-autocast emulation (OracleNeRF.rgba(autocast_half=True), train_step(scaler=...)) - the
    thing the unchanged trainer's 1000-step PSNR is compared with - pinned to the reference's own modules: OctreeAS -> HashGrid
    (ops/grid.py's HashGridInterpolate with its `codebook.half()` branch taken) -> NeuralRadianceField (decoders, embedder, rgba
    body) -> PackedRFTracer, executed where they lie under torch.autocast('cpu', dtype=torch.float16), the loss scaled by
    GradScaler's initial 65536 and back-propagated inside the autocast region like BaseTrainer.iterate does (base_trainer.py:338).
    CPU autocast applies the same cast policy as CUDA autocast to the ops on this path (linear -> fp16; relu / sigmoid / cat
    follow their inputs; the positional embedding stays fp32 and `cat` promotes), so every rounding point of the reference is
    exercised.  Pinned: per-sample colours and densities (bit for bit), rendered rgb, loss, every decoder gradient.  NOT pinned, by
    design: the table gradient - the reference rounds every __half2 atomicAdd to fp16 in arrival order
    (hashgrid_interpolate_cuda.cu:133-160), the oracle (like the HIP path) accumulates the same fp16 upstream gradient in fp32;
    the two differ by at most one fp16 rounding of the sum here (the host build of the kernel body adds in float and rounds once)."""
    from oracle import nerf as onerf
    from wisp.core import Rays
    t = torch.from_numpy
    blas_mod, RefGrid, RefField, RefTracer = _reference_nerf_stack(monkeypatch, half_autocast=True)
    rng = np.random.default_rng(191)
    pts = rng.integers(0, 16, size=(600, 3))
    steps, R, bg, scale = 64, 96, (0.0, 0.0, 0.0), 65536.0
    blas = blas_mod["OctreeAS"].from_quantized_points(t(pts.astype(np.int16)), 4)
    torch.manual_seed(192)
    grid = RefGrid.from_geometric(blas, feature_dim=2, num_lods=4, multiscale_type='cat', feature_std=0.3, codebook_bitwidth=10,
                                  min_grid_res=8, max_grid_res=64)
    nef = RefField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1, bias=True)
    tracer = RefTracer(raymarch_type='ray', num_steps=steps, bg_color=bg)
    onef = onerf.OracleNeRF([int(r) for r in grid.resolutions], 2, 10, 'cat', 0.3, 64, 1, True, 4)
    onef.load_state_dict(nef.state_dict(), strict=False)
    oblas = onerf.OracleBLAS.from_quantized_points(pts, 4)
    o = rng.normal(size=(R, 3)).astype(np.float32)
    o = 3.0 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = -o + rng.normal(size=o.shape).astype(np.float32) * 0.4
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    gts = t(rng.uniform(0, 1, (R, 3)).astype(np.float32))
    jit = rng.uniform(size=(R, steps)).astype(np.float32)
    blas_mod["torch"] = _TorchWithDraws(t(jit))

    # ---- the reference's modules under fp16 autocast; per-sample outputs tapped at the field's rgba
    tapped = {}
    real_rgba = nef.rgba
    def rgba(self, coords, ray_d, lod_idx=None):                 # (the dispatcher fills a bound method's arguments by signature)
        out = real_rgba(coords, ray_d, lod_idx=lod_idx)
        tapped["rgb"], tapped["density"] = out["rgb"].detach().clone(), out["density"].detach().clone()
        return out
    rgba = types.MethodType(rgba, nef)
    swapped = {(rgba if getattr(f, "__func__", None) is real_rgba.__func__ else f): ch for f, ch in nef._forward_functions.items()}
    assert rgba in swapped
    nef._forward_functions = swapped
    with torch.autocast('cpu', dtype=torch.float16):
        rb = tracer(nef, rays=Rays(t(o), t(d), dist_min=1.0, dist_max=5.0), channels=["rgb"])
        loss = torch.nn.functional.smooth_l1_loss(rb.rgb, gts, reduction='none').mean()
        (loss * scale).backward()
    assert tapped and tapped["rgb"].dtype == torch.float16 and tapped["density"].dtype == torch.float16    # autocast really was on
    assert rb.rgb.dtype == torch.float32                                                                   # compositing promotes

    # ---- the oracle's emulation
    res = onerf.trace(onef, oblas, t(o), t(d), 1.0, 5.0, steps, jit, bg, 'ray', with_depth=False, autocast_half=True)
    want_loss = torch.nn.functional.smooth_l1_loss(res["rgb"], gts, reduction='none').mean()
    (want_loss * scale).backward()
    assert tracer.prev_num_samples == res["raymarch"]["ridx"].shape[0] > 300
    assert torch.equal(res["sample_rgb"], tapped["rgb"].float()), float((res["sample_rgb"] - tapped["rgb"].float()).abs().max())
    assert torch.equal(res["sample_density"], tapped["density"].float())
    assert float((res["sample_rgb"] - res["sample_rgb"].half().float()).abs().max()) == 0.0               # they ARE fp16 values
    assert torch.allclose(rb.rgb, res["rgb"], atol=1e-6, rtol=0)
    assert abs(float(loss) - float(want_loss)) < 1e-7
    # the fp32 oracle is a different function: this test would notice the emulation being switched off
    with torch.no_grad():
        plain = onerf.trace(onef, oblas, t(o), t(d), 1.0, 5.0, steps, jit, bg, 'ray', with_depth=False)
    assert float((plain["rgb"] - res["rgb"]).abs().max()) > 1e-5
    theirs = dict(onef.named_parameters())
    seen, differing = 0, []
    for n, p in nef.named_parameters():
        if not p.requires_grad:
            continue
        g_ref, g_ora = p.grad, theirs[n].grad
        assert g_ref is not None and g_ora is not None and g_ref.dtype == g_ora.dtype == torch.float32, n
        sc = float(g_ref.abs().max())
        assert sc > 0 and bool(torch.isfinite(g_ref).all())
        err = float((g_ref - g_ora).abs().max())
        if "grid" in n:
            # one fp16 rounding of each entry's sum on the reference side (2^-11 relative to the entry), see the docstring
            # (2^-11 of the entry, or half the fp16 subnormal step 2^-24 for entries below fp16's normal range; gradients carry the loss scale)
            # + the two fp32 sums' add-order noise, 1e-6 of the tensor's largest entry (the bound the decoder gradients get below)
            worst = float(((g_ref - g_ora).abs() / (g_ora.abs() * 2.0 ** -11 + 2.0 ** -25 + 1e-6 * sc)).max())
            assert worst <= 1.0, (n, err, sc, worst)
            assert err > 0                                     # (the reference side really went through fp16)
        else:
            # decoder gradients are fp16 values on both sides (autocast's weight casts): the two compositing graphs order their fp32
            # operations differently (1e-7 relative), which now and then tips an fp16 rounding of the upstream gradient - entries
            # differ by one fp16 ulp of themselves, or (sums that cancel) by one tipped upstream element's share: 3e-5 of the tensor's
            # largest entry (measured 9e-6), and few entries differ at all
            ulp = g_ora.abs() * 2.0 ** -10 + 2.0 ** -24 + 3e-5 * sc
            assert bool(((g_ref - g_ora).abs() <= ulp).all()), (n, err, sc)
            differing.append((n, float((g_ref != g_ora).float().mean())))
        seen += 1
    assert seen == 11 and len(differing) == 10
    assert max(f for _, f in differing) <= 0.10 and sum(f for _, f in differing) / 10 <= 0.03, differing     # measured 0.055 / 0.018


def test_oracle_sdf_render_equals_the_whole_reference_stack_on_the_host(monkeypatch):
    """The C3 path end to end on the CPU with the reference's OWN classes: OctreeAS -> OctreeGrid (models/grids/octree_grid.py) ->
    NeuralSDF (models/nefs/neural_sdf.py) -> PackedSDFTracer (tracers/packed_sdf_tracer.py) with the reference's find_depth_bound
    wrapper over its kernel body built for the host, every module executed where it lies, only the Kaolin leaves supplied - against
    oracle.sdf.sphere_trace over the oracle's octree-grid lookup and decoder with the same parameters.  The decoder is set to an
    octahedron-like distance (0.6 |x|_1 - 0.5 from ReLU pairs) plus small grid-feature terms, the occupancy is a shell of level-5 cells
    around that surface, so rays hit, graze and miss."""
    from oracle import nerf as onerf, octree_grid as og, sdf as osdf, spc as ospc, ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built")
    import wisp.ops.geometric as geometric
    from wisp.core import Rays
    t = torch.from_numpy
    stubs = _kaolin_stub()
    kspc = stubs["kaolin.ops.spc"]
    kspc.unbatched_points_to_octree = lambda p, level, sorted=False: t(ospc.points_to_octree(p.cpu().numpy(), level))
    kspc.unbatched_query = lambda octree, prefix, coords, level, with_parents=False: t(
        ospc.query(octree.numpy(), prefix.numpy(), coords.detach().numpy(), level, with_parents=with_parents))
    kspc.unbatched_interpolate_trilinear = lambda c, pidx, pts, tr, f, lod: og.interpolate_trilinear(c, pidx.long(), pts, tr, f.float(), lod,
                                                                                                      half_round=True)
    krender, krs = types.ModuleType("kaolin.render"), types.ModuleType("kaolin.render.spc")
    krs.mark_pack_boundaries = lambda ridx: t(ospc.mark_pack_boundaries(ridx.numpy()))

    def raytrace(octree, points, pyramid, prefix, origins, dirs, level, return_depth=True, with_exit=False):
        ridx, pidx, depth = ospc.raytrace(octree.numpy(), points.numpy(), pyramid.numpy(), prefix.numpy(), origins.numpy(), dirs.numpy(),
                                          level, with_exit=with_exit)
        return t(ridx), t(pidx), t(depth.copy())
    krs.unbatched_raytrace = raytrace
    krender.spc = krs
    stubs["kaolin"].render = krender
    stubs.update({"kaolin.render": krender, "kaolin.render.spc": krs})
    for name, mod in stubs.items():
        monkeypatch.setitem(sys.modules, name, mod)
    c_ext = types.SimpleNamespace(render=types.SimpleNamespace(find_depth_bound_cuda=lambda q, cur, dep: t(
        ref_lib.find_depth_bound(q.numpy(), cur.numpy(), dep.numpy()))))
    monkeypatch.setattr(geometric, "find_depth_bound",
                        _reference_function("ops/geometric.py", "find_depth_bound", dict(torch=torch, _C=c_ext)))
    RefAS = _exec_reference("accelstructs/octree_as.py")["OctreeAS"]
    RefGrid = _exec_reference("models/grids/octree_grid.py")["OctreeGrid"]
    RefField = _exec_reference("models/nefs/neural_sdf.py")["NeuralSDF"]
    RefTracer = _exec_reference("tracers/packed_sdf_tracer.py")["PackedSDFTracer"]

    level, F = 5, 4
    cells = np.stack(np.meshgrid(*[np.arange(32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    centre = (cells + 0.5) / 16.0 - 1.0
    shell = cells[np.abs(0.6 * np.abs(centre).sum(1) - 0.5) < 0.08]
    blas = RefAS.from_quantized_points(t(shell.astype(np.int16)), level)
    torch.manual_seed(95)
    grid = RefGrid(blas, feature_dim=F, num_lods=3, interpolation_type='linear', multiscale_type='sum', feature_std=0.05)
    nef = RefField(grid, pos_embedder='none', position_input=True, hidden_dim=32, num_layers=1)
    assert grid.active_lods == [3, 4, 5] and nef.decoder.layers[0].weight.shape == (32, 3 + F)
    with torch.no_grad():                                            # hidden 0..5 = relu(+-x_i): 0.6 |x|_1 - 0.5; the rest: small feature terms
        w1, b1, w2, b2 = nef.decoder.layers[0].weight, nef.decoder.layers[0].bias, nef.decoder.lout.weight, nef.decoder.lout.bias
        w1[:6].zero_(); b1.zero_(); w1[6:, :3].zero_()
        for i in range(3):
            w1[2 * i, i], w1[2 * i + 1, i] = 1.0, -1.0
        w2[0, :6] = 0.6
        w2[0, 6:] *= 0.05
        b2.fill_(-0.5)
    tracer = RefTracer(num_steps=40, step_size=0.8, min_dis=3e-4)
    rng = np.random.default_rng(96)
    o = rng.normal(size=(150, 3)).astype(np.float32)
    o = (2.5 * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    d = -o + rng.normal(size=o.shape).astype(np.float32) * 0.5
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rb = tracer(nef, rays=Rays(t(o), t(d), dist_min=0.0, dist_max=6.0), channels={"rgb", "normal", "depth", "hit"})

    oblas = onerf.OracleBLAS.from_quantized_points(shell, level)
    assert np.array_equal(oblas.octree, blas.octree.numpy())
    pd, pyd = ospc.make_dual(oblas.points, oblas.pyramid)
    trinkets, _ = ospc.make_trinkets(oblas.points, oblas.pyramid, pd, pyd)
    assert np.array_equal(np.asarray(trinkets), grid.trinkets.numpy())
    dec = onerf.OracleDecoder(3 + F, 1, 32, 1, True)
    dec.load_state_dict(nef.decoder.state_dict())
    feats = [f.detach() for f in grid.features]

    def field(x):
        with torch.no_grad():
            f = og.octree_grid_interpolate(oblas, trinkets, feats, x, 2, grid.base_lod, grid.active_lods, 'sum', F, half_round=True)
            return dec(torch.cat([x, f], -1))
    want = osdf.sphere_trace(field, oblas, t(o), t(d), 6.0, level, 40, 0.8, 3e-4)
    hits = int(rb.hit.sum())
    assert torch.equal(rb.hit, want["hit"]) and 25 < hits < 150, hits
    for name in ("xyz", "depth", "normal", "rgb", "alpha"):
        got, ref = getattr(rb, name), want[name]
        assert got.shape == ref.shape and torch.allclose(got, ref, atol=2e-6, rtol=0), (name, float((got - ref).abs().max()))
    on_surface = field(rb.xyz[rb.hit]).abs()
    assert float(on_surface.max()) < 5e-3                             # the hits really sit on the zero level of the field


def test_oracle_image_fit_equals_the_whole_reference_stack_on_the_host(monkeypatch):
    """C1 (app/image, BASELINE configs[0]: the reference's own CPU-runnable case) with the reference's OWN classes on the CPU:
    HashGrid.from_geometric(blas=None) over its ops/grid.py and its 2-D (bilinear) hash-grid kernel bodies built for the host ->
    ImageNeuralField (models/nefs/image_nef.py), executed where they lie - against the oracle composition the GPU C1 test uses,
    sigmoid(decoder(cat([2-D grid features, 3-octave embedding of the pixel coordinate]))): the forward over a 48 x 48 image and five Adam
    steps of the fit (loss at every step, parameters afterwards)."""
    from oracle import hashgrid as ohash, nerf as onerf
    t = torch.from_numpy
    _, RefGrid, _, _ = _reference_nerf_stack(monkeypatch)
    RefImage = _exec_reference("models/nefs/image_nef.py")["ImageNeuralField"]
    torch.manual_seed(101)
    grid = RefGrid.from_geometric(None, feature_dim=2, num_lods=8, multiscale_type='cat', feature_std=0.05, codebook_bitwidth=12,
                                  min_grid_res=8, max_grid_res=64)
    nef = RefImage(grid, hidden_dim=32)
    res = [int(r) for r in grid.resolutions]
    assert nef.input_dim == 2 * 8 + 14 and grid.codebook.feats.shape[0] == sum(min(2 ** 12, r ** 3) for r in res)   # table sized for 3-D
    H = 48
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, H), indexing='ij')
    coords = torch.stack([xs, ys], -1).reshape(-1, 2)
    img = torch.stack([0.5 + 0.5 * torch.sin(6 * xs), 0.5 + 0.5 * torch.cos(4 * ys), 0.5 + 0.5 * torch.sin(5 * xs * ys)], -1).reshape(-1, 3)

    table = grid.codebook.feats.detach().clone().requires_grad_(True)
    dec = onerf.OracleDecoder(nef.input_dim, 3, 32, 1, True)
    dec.load_state_dict(nef.decoder.state_dict())

    def oracle_rgb(x):
        feats = ohash.grid_interpolate(x, len(res) - 1, 'cat', 2, res, 12, table, grid.codebook.begin_idxes)
        return torch.sigmoid(dec(torch.cat([feats, onerf.positional_embed(x, 3, include_input=True)], -1)))

    with torch.no_grad():
        got, want = nef.rgb(coords), oracle_rgb(coords)          # called directly, as app/image does: rgb() returns a bare tensor, which
                                                                 # BaseNeuralField.forward cannot index by channel (upstream too)
    assert got.shape == (H * H, 3) and torch.allclose(got, want, atol=1e-6, rtol=0) and float(got.std()) > 1e-3
    opt_ref = torch.optim.Adam([{"params": grid.parameters(), "lr": 0.05}, {"params": nef.decoder.parameters(), "lr": 1e-3}], eps=1e-15)
    opt_ora = torch.optim.Adam([{"params": [table], "lr": 0.05}, {"params": dec.parameters(), "lr": 1e-3}], eps=1e-15)
    g = torch.Generator().manual_seed(102)
    losses = []
    for step in range(5):
        idx = torch.randint(0, coords.shape[0], (1024,), generator=g)
        la = ((nef.rgb(coords[idx]) - img[idx]) ** 2).mean()
        opt_ref.zero_grad(); la.backward(); opt_ref.step()
        lb = ((oracle_rgb(coords[idx]) - img[idx]) ** 2).mean()
        opt_ora.zero_grad(); lb.backward(); opt_ora.step()
        assert abs(float(la) - float(lb)) < 2e-6, (step, float(la), float(lb))
        losses.append(float(la))
    assert losses[-1] < losses[0]
    touched = (grid.codebook.feats.detach() - table.detach()).abs()
    assert float((touched > 2e-5).float().mean()) < 1e-3 and float(touched.max()) < 2e-3      # Adam on noise-level gradients, see above
    for p, q in zip(nef.decoder.parameters(), dec.parameters()):
        assert torch.allclose(p, q, atol=1e-5, rtol=0)


def test_oracle_sdf_training_equals_the_whole_reference_stack_on_the_host(monkeypatch):
    """C3 training on the CPU with the reference's OWN classes: NeuralSDF over OctreeGrid over OctreeAS, executed where they lie; the loss
    of SDFTrainer.step (trainers/sdf_trainer.py:96-102: sum over the loss LODs of sum((pred - gts)^2)) over every level index, the
    gradient flowing through OctreeGrid.interpolate / _interpolate into each level's feature table; Adam as nglod_octree.yaml sets it -
    next to the oracle composition decoder(cat([position, oracle octree-grid lookup])) from the same state and batches: the same loss at
    every step, the same gradient for every table and decoder tensor at the first step, the same parameters after four."""
    from oracle import nerf as onerf, octree_grid as og, spc as ospc
    t = torch.from_numpy
    stubs = _kaolin_stub()
    kspc = stubs["kaolin.ops.spc"]
    kspc.unbatched_points_to_octree = lambda p, level, sorted=False: t(ospc.points_to_octree(p.cpu().numpy(), level))
    kspc.unbatched_query = lambda octree, prefix, coords, level, with_parents=False: t(
        ospc.query(octree.numpy(), prefix.numpy(), coords.detach().numpy(), level, with_parents=with_parents))
    kspc.unbatched_interpolate_trilinear = lambda c, pidx, pts, tr, f, lod: og.interpolate_trilinear(c, pidx.long(), pts, tr, f.float(), lod,
                                                                                                      half_round=True)
    krender, krs = types.ModuleType("kaolin.render"), types.ModuleType("kaolin.render.spc")
    krender.spc = krs
    stubs["kaolin"].render = krender
    stubs.update({"kaolin.render": krender, "kaolin.render.spc": krs})
    for name, mod in stubs.items():
        monkeypatch.setitem(sys.modules, name, mod)
    RefAS = _exec_reference("accelstructs/octree_as.py")["OctreeAS"]
    RefGrid = _exec_reference("models/grids/octree_grid.py")["OctreeGrid"]
    RefField = _exec_reference("models/nefs/neural_sdf.py")["NeuralSDF"]
    rng = np.random.default_rng(111)
    level, F, lods = 5, 4, 3
    cells = rng.integers(0, 32, size=(900, 3))
    blas = RefAS.from_quantized_points(t(cells.astype(np.int16)), level)
    torch.manual_seed(112)
    grid = RefGrid(blas, feature_dim=F, num_lods=lods, interpolation_type='linear', multiscale_type='sum', feature_std=0.1)
    nef = RefField(grid, pos_embedder='none', position_input=True, hidden_dim=32, num_layers=1)
    oblas = onerf.OracleBLAS.from_quantized_points(cells, level)
    pd, pyd = ospc.make_dual(oblas.points, oblas.pyramid)
    trinkets, _ = ospc.make_trinkets(oblas.points, oblas.pyramid, pd, pyd)
    tables = [f.detach().clone().requires_grad_(True) for f in grid.features]
    dec = onerf.OracleDecoder(3 + F, 1, 32, 1, True)
    dec.load_state_dict(nef.decoder.state_dict())

    def oracle_sdf(x, lod_idx):
        f = og.octree_grid_interpolate(oblas, trinkets, tables, x, lod_idx, grid.base_lod, grid.active_lods, 'sum', F, half_round=True)
        return dec(torch.cat([x, f], -1))

    opt_ref = torch.optim.Adam(nef.parameters(), lr=1e-3, eps=1e-15)                # nglod_octree.yaml: adam, lr 1e-3, eps 1e-15
    opt_ora = torch.optim.Adam(tables + list(dec.parameters()), lr=1e-3, eps=1e-15)
    leaf = (oblas.level_points().astype(np.float32) + 0.5) / 16.0 - 1.0            # sample inside occupied cells (+ a few outside)
    for step in range(4):
        pick = leaf[rng.integers(0, leaf.shape[0], 256)] + rng.uniform(-0.03, 0.03, (256, 3)).astype(np.float32)
        pts = t(np.concatenate([pick, rng.uniform(-1, 1, (32, 3)).astype(np.float32)]).astype(np.float32))
        gts = t(rng.normal(size=(288, 1)).astype(np.float32) * 0.1)
        la = sum(((nef(coords=pts, lod_idx=i, channels="sdf") - gts) ** 2).sum() for i in range(lods))
        opt_ref.zero_grad(); la.backward()
        lb = sum(((oracle_sdf(pts, i) - gts) ** 2).sum() for i in range(lods))
        opt_ora.zero_grad(); lb.backward()
        assert abs(float(la) - float(lb)) < 2e-5 * max(1.0, abs(float(lb))), (step, float(la), float(lb))
        if step == 0:
            for i, (a, b) in enumerate(zip(grid.features, tables)):
                scale = float(b.grad.abs().max())
                assert scale > 0 and torch.allclose(a.grad, b.grad, atol=2e-5 * scale, rtol=0), (i, float((a.grad - b.grad).abs().max()), scale)
            for a, b in zip(nef.decoder.parameters(), dec.parameters()):
                assert torch.allclose(a.grad, b.grad, atol=2e-5 * float(b.grad.abs().max()) + 1e-9, rtol=0)
        opt_ref.step(); opt_ora.step()
    for a, b in zip(list(grid.features) + list(nef.decoder.parameters()), tables + list(dec.parameters())):
        diff = (a.detach() - b.detach()).abs()
        assert float((diff > 2e-6).float().mean()) < 2e-3 and float(diff.max()) < 5e-4, float(diff.max())


@pytest.mark.parametrize("training", [False, True])
def test_oracle_codebook_render_equals_the_whole_reference_stack_on_the_host(monkeypatch, training):
    """C5 (VQAD) end to end on the CPU with the reference's OWN classes: OctreeAS -> CodebookOctreeGrid (models/grids/codebook_grid.py,
    with the OctreeGrid it derives from) -> NeuralRadianceField without biases (nerf_codebook.yaml) -> PackedRFTracer, 'voxel' march at
    the grid's base level, executed where they lie; only the Kaolin leaves and the jitter draw supplied.  Against oracle.nerf.trace over
    an adapter field built from oracle.octree_grid.codebook_grid_interpolate and the oracle decoders: eval mode (argmax rows of the
    dictionary) and training mode (straight-through softmax keys)."""
    from oracle import nerf as onerf, octree_grid as og, spc as ospc
    from wisp.core import Rays
    import wisp.ops.spc as package_spc
    t = torch.from_numpy
    blas_mod, _, RefField, RefTracer = _reference_nerf_stack(monkeypatch)
    sys.modules["kaolin.ops.spc"].coords_to_trilinear_coeffs = lambda c, pts, lod: og.trilinear_coeffs(c, pts.long(), lod)
    _exec_reference("models/grids/octree_grid.py")                                  # registers nothing; the codebook module imports the package's
    RefGrid = _exec_reference("models/grids/codebook_grid.py")["CodebookOctreeGrid"]
    rng = np.random.default_rng(121)
    level, F, lods, K, steps, R, bg = 5, 5, 3, 16, 4, 130, (1.0, 1.0, 1.0)
    cells = rng.integers(0, 32, size=(2500, 3))
    blas = blas_mod["OctreeAS"].from_quantized_points(t(cells.astype(np.int16)), level)
    torch.manual_seed(122)
    grid = RefGrid(blas, feature_dim=F, num_lods=lods, interpolation_type='linear', multiscale_type='sum', feature_std=0.5,
                   codebook_bitwidth=4)
    nef = RefField(grid, view_embedder='positional', view_multires=4, hidden_dim=32, num_layers=1, bias=False)
    nef.train(training)
    assert grid.base_lod == 3 and grid.active_lods == [3, 4, 5] and nef.decoder_density.lout.bias is None
    tracer = RefTracer(raymarch_type='voxel', num_steps=steps, bg_color=bg)
    o = rng.normal(size=(R, 3)).astype(np.float32)
    o = 3.0 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = -o + rng.normal(size=o.shape).astype(np.float32) * 0.4
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    oblas = onerf.OracleBLAS.from_quantized_points(cells, level)
    nuggets = ospc.raytrace(oblas.octree, oblas.points, oblas.pyramid, oblas.exsum, o, d, grid.base_lod, True)[0].shape[0]
    jit = rng.uniform(size=(nuggets, steps)).astype(np.float32)
    sampling = _exec_reference("ops/spc/sampling.py")
    sampling["torch"] = _TorchWithDraws(t(jit))

    class SpcOps:
        sample_from_depth_intervals = staticmethod(sampling["sample_from_depth_intervals"])
        expand_pack_boundary = staticmethod(sampling["expand_pack_boundary"])

        def __getattr__(self, name):
            return getattr(package_spc, name)
    blas_mod["wisp_spc_ops"] = SpcOps()
    with torch.no_grad():
        rb = tracer(nef, rays=Rays(t(o), t(d), dist_min=1.0, dist_max=5.0), channels={"rgb", "depth", "alpha", "hit"})

    # ---- oracle: the same arithmetic after the codebook lookup as OracleNeRF.rgba (nerf.py:245-264)
    pd, pyd = ospc.make_dual(oblas.points, oblas.pyramid)
    trinkets, _ = ospc.make_trinkets(oblas.points, oblas.pyramid, pd, pyd)
    dd, dc = onerf.OracleDecoder(F, 16, 32, 1, False), onerf.OracleDecoder(15 + nef.view_embed_dim, 3, 32, 2, False)
    dd.load_state_dict(nef.decoder_density.state_dict())
    dc.load_state_dict(nef.decoder_color.state_dict())
    logits, dictionary = [f.detach() for f in grid.features], [x.detach() for x in grid.dictionary]

    class Field:
        @staticmethod
        def rgba(coords, ray_d, lod_idx=None):
            lod_idx = lods - 1 if lod_idx is None else lod_idx
            feats = og.codebook_grid_interpolate(oblas, trinkets, logits, dictionary, coords, lod_idx, grid.active_lods, 'sum', F, training)
            y = dd(feats)
            rgb = torch.sigmoid(dc(torch.cat([y, onerf.positional_embed(ray_d, 4, include_input=True)], -1)[..., 1:]))
            return dict(rgb=rgb, density=torch.relu(y[..., 0:1]))

    march_view = types.SimpleNamespace(octree=oblas.octree, points=oblas.points, pyramid=oblas.pyramid, exsum=oblas.exsum,
                                       max_level=grid.base_lod)                      # OctreeGrid.raymarch marches at base_lod (:221-226)
    with torch.no_grad():
        want = onerf.trace(Field, march_view, t(o), t(d), 1.0, 5.0, steps, jit, bg, 'voxel', with_depth=True)
    assert tracer.prev_num_samples == want["raymarch"]["ridx"].shape[0] == nuggets * steps > 800
    assert torch.equal(rb.hit, want["hit"]) and 20 < int(rb.hit.sum()) <= R
    for name, rtol in (("rgb", 0.0), ("alpha", 0.0), ("depth", 2e-6)):
        got, ref = getattr(rb, name), want[name]
        assert got.shape == ref.shape and torch.allclose(got, ref, atol=3e-6, rtol=rtol), (name, float((got - ref).abs().max()))
    assert float(rb.rgb.std()) > 0.02


def test_oracle_prune_then_render_equals_the_whole_reference_stack_on_the_host(monkeypatch):
    """NeuralRadianceField.prune (models/nefs/nerf.py:175-212) as a METHOD of the reference's own field over its own HashGrid and
    OctreeAS on the CPU (draws injected): occupancy decay + max, threshold, and the BLAS replaced through
    `blas.__class__.from_quantized_points` - then a render of the pruned scene through the reference's tracer.  Against
    oracle.nerf.prune followed by oracle.nerf.trace on the pruned oracle BLAS: same occupancy record, same octree, same image."""
    from oracle import nerf as onerf
    from wisp.core import Rays
    t = torch.from_numpy
    blas_mod, RefGrid, RefField, RefTracer = _reference_nerf_stack(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    nerf_mod = sys.modules[RefField.__module__].__dict__
    nerf_mod["HashGrid"] = RefGrid                                                  # its isinstance(self.grid, (HashGrid, TriplanarGrid))
    level, steps, R, bg = 4, 80, 120, (0.1, 0.2, 0.3)
    blas = blas_mod["OctreeAS"].make_dense(level)
    torch.manual_seed(131)
    grid = RefGrid.from_geometric(blas, feature_dim=2, num_lods=4, multiscale_type='cat', feature_std=0.5, codebook_bitwidth=10,
                                  min_grid_res=8, max_grid_res=64)
    nef = RefField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1, bias=True,
                   prune_density_decay=0.95, prune_min_density=1.0)
    onef = onerf.OracleNeRF([int(r) for r in grid.resolutions], 2, 10, 'cat', 0.5, 64, 1, True, 4)
    onef.load_state_dict(nef.state_dict(), strict=False)
    oblas = onerf.OracleBLAS.make_dense(level)
    cells = oblas.level_points()
    assert np.array_equal(grid.dense_points.numpy(), cells) and grid.occupancy.shape == (16 ** 3,)
    rng = np.random.default_rng(132)
    unit = t(rng.uniform(size=(cells.shape[0], 3)).astype(np.float32))
    views = rng.normal(size=(cells.shape[0], 3)).astype(np.float32)
    views /= np.linalg.norm(views, axis=1, keepdims=True)
    with torch.no_grad():                                                            # a threshold that prunes about half of the cells
        probe = onef.rgba(((t(cells.astype(np.float32)) + unit) / 16.0) * 2.0 - 1.0, t(views))["density"][:, 0]
    nef.prune_min_density = float(probe.median())
    nerf_mod["torch"] = _TorchWithDraws(unit)                                        # prune's torch.rand(points.shape[0], 3)
    nerf_mod["sample_unif_sphere"] = lambda n: views
    nef.prune()
    new_oblas, occupancy = onerf.prune(onef, oblas, torch.zeros(cells.shape[0]), cells, 0.95, nef.prune_min_density, unit, t(views))
    kept = int(grid.blas.pyramid[0, level])
    assert 0.3 * 4096 < kept < 0.7 * 4096 and isinstance(grid.blas, blas_mod["OctreeAS"]) and grid.blas is not blas
    assert torch.allclose(grid.occupancy, occupancy, atol=1e-6, rtol=0)
    assert np.array_equal(grid.blas.octree.numpy(), new_oblas.octree)
    nef.prune()                                                                      # second round: the decayed record matters now
    new_oblas2, occupancy2 = onerf.prune(onef, new_oblas, occupancy, cells, 0.95, nef.prune_min_density, unit, t(views))
    assert torch.allclose(grid.occupancy, occupancy2, atol=1e-6, rtol=0) and np.array_equal(grid.blas.octree.numpy(), new_oblas2.octree)

    nerf_mod["torch"] = torch
    tracer = RefTracer(raymarch_type='ray', num_steps=steps, bg_color=bg)
    o = rng.normal(size=(R, 3)).astype(np.float32)
    o = 3.0 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = -o + rng.normal(size=o.shape).astype(np.float32) * 0.3
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    jit = rng.uniform(size=(R, steps)).astype(np.float32)
    blas_mod["torch"] = _TorchWithDraws(t(jit))
    with torch.no_grad():
        rb = tracer(nef, rays=Rays(t(o), t(d), dist_min=1.0, dist_max=5.0), channels={"rgb", "alpha", "hit"})
        want = onerf.trace(onef, new_oblas2, t(o), t(d), 1.0, 5.0, steps, jit, bg, 'ray', with_depth=False)
    assert tracer.prev_num_samples == want["raymarch"]["ridx"].shape[0] > 500
    assert np.array_equal(rb.hit.numpy(), want["hit"].numpy()) and int(rb.hit.sum()) > 100
    # ~23 samples per ray in this dense scene and deltas down to 3e-3: a last-bit difference of a sample depth near 3 (torch's CPU
    # arithmetic vs the GPU-order arithmetic the oracle restates, see the raymarch pin) is 1e-4 of such a delta and adds up through
    # tau = density x delta along the ray - measured 1.2e-5 on alpha, 5e-6 on rgb
    assert torch.allclose(rb.rgb, want["rgb"], atol=3e-5, rtol=0) and torch.allclose(rb.alpha, want["alpha"], atol=3e-5, rtol=0)
    assert float(want["alpha"].max() - want["alpha"].min()) > 0.3


def test_differential_helpers_equal_the_reference_function_bodies():
    """finitediff_gradient, tetrahedron_gradient and autodiff_gradient (ops/differential/gradients.py:14-95) compiled from the reference
    file against wisp.ops.differential on an analytic field: identical estimates, all three close to the true gradient, and
    autodiff_gradient differentiable a second time (create_graph)."""
    import wisp.ops.differential as mine
    glb = dict(torch=torch)
    field = lambda p: (p[..., 0:1] ** 2) * 0.5 + torch.sin(3.0 * p[..., 1:2]) * 0.2 + p[..., 2:3] * p[..., 0:1]
    truth = lambda p: torch.cat([p[..., 0:1] + p[..., 2:3], 0.6 * torch.cos(3.0 * p[..., 1:2]), p[..., 0:1]], -1)
    torch.manual_seed(141)
    x = torch.rand(200, 3) * 2 - 1
    for name, kw, tol in (("finitediff_gradient", dict(eps=0.005), 1e-4), ("tetrahedron_gradient", dict(eps=0.005), 2e-2),
                          ("autodiff_gradient", {}, 1e-6)):
        ref = _reference_function("ops/differential/gradients.py", name, glb)
        a, b = ref(x.clone(), field, **kw), getattr(mine, name)(x.clone(), field, **kw)
        assert a.shape == b.shape == (200, 3) and torch.allclose(a, b, atol=2e-6, rtol=0), (name, float((a - b).abs().max()))
        assert float((b.detach() - truth(x)).abs().max()) < tol, name
    xx = x.clone()
    first = mine.autodiff_gradient(xx, field)
    assert first.requires_grad                                                     # the graph is kept: differentiable once more
    second = torch.autograd.grad(first[:, 0].sum(), xx)[0]                         # d/dp of (x + z) = (1, 0, 1)
    assert torch.allclose(second, torch.tensor([1.0, 0.0, 1.0]).expand(200, 3), atol=1e-6)


def test_profiler_ranges_are_entered_at_the_sites_the_reference_marks(monkeypatch):
    """VERDICT r4 missing-2: the reference brackets its trainer with profiler ranges - `MultiviewTrainer.step`
    (multiview_trainer.py:111), `MultiviewTrainer.backward` (:169), `SampleRays` (ray_sampler.py:24), and torch's per-op ranges
    around the whole run (base_trainer.py:368, cfg.profile_nvtx).  This package enters ranges of the same names at the same places
    (torch.cuda.nvtx is roctx on ROCm), in the unchanged-trainer class and in the fused step, balanced push / pop."""
    import test_distributed_gloo as stub
    import wisp._C as C
    from wisp.core import Rays
    from wisp.datasets.batch import MultiviewBatch
    from wisp.datasets.transforms import SampleRays
    from wisp.trainers import MultiviewTrainer, ConfigMultiviewTrainer, ConfigAdamW, MultiviewTrainStep
    # the names, read from the reference's own sources
    src = {f: open(os.path.join(REF, f)).read() for f in ("trainers/multiview_trainer.py", "datasets/transforms/ray_sampler.py",
                                                                 "trainers/base_trainer.py")}
    assert '@torch.cuda.nvtx.range("MultiviewTrainer.step")' in src["trainers/multiview_trainer.py"]
    assert 'torch.cuda.nvtx.range("MultiviewTrainer.backward")' in src["trainers/multiview_trainer.py"]
    assert '@torch.cuda.nvtx.range("SampleRays")' in src["datasets/transforms/ray_sampler.py"]
    assert "emit_nvtx(enabled=self.cfg.profile_nvtx)" in src["trainers/base_trainer.py"]
    log, depth = [], [0]
    monkeypatch.setattr(torch.cuda.nvtx, "range_push", lambda msg: (log.append(msg), depth.__setitem__(0, depth[0] + 1)))
    monkeypatch.setattr(torch.cuda.nvtx, "range_pop", lambda: depth.__setitem__(0, depth[0] - 1))
    emitted = []
    real_emit = torch.autograd.profiler.emit_nvtx
    monkeypatch.setattr(torch.autograd.profiler, "emit_nvtx", lambda enabled=True, **k: (emitted.append(enabled), real_emit(enabled=False))[1])
    g = torch.Generator().manual_seed(3)
    O, D, T = torch.rand(64, 3, generator=g) * 2 - 1, torch.randn(64, 3, generator=g), torch.rand(64, 3, generator=g)
    # SampleRays
    out = SampleRays(16)(MultiviewBatch(rays=Rays(O, D), rgb=T))
    assert out["rays"].origins.shape == (16, 3) and log == ["SampleRays"] and depth[0] == 0
    # the unchanged trainer: train() -> iterate() -> step()
    pipe = stub._StubPipeline()
    pipe.tracer.raymarch_type, pipe.tracer.num_steps, pipe.tracer.prev_num_samples = 'ray', 64, None
    pipe.nef.grid.raymarch = lambda rays, **kw: types.SimpleNamespace(samples=torch.zeros(64 * 7, 3))
    pipe.nef.grid.active_lods = [0]

    class _Views:
        def __init__(self):
            self.transform = SampleRays(64)
        def __len__(self):
            return 3
    class _Loader:
        def __len__(self):
            return 3
        def __iter__(self):
            return iter([{"rays": Rays(O[None], D[None]), "rgb": T[None]} for _ in range(3)])
    cfg = ConfigMultiviewTrainer(optimizer=ConfigAdamW(lr=1e-2, eps=1e-16, weight_decay=1e-6), grid_lr_weight=10.0, max_epochs=1,
                                 enable_amp=False, prune_every=-1, rgb_loss_type='huber', target_sample_size=2 ** 12)
    assert cfg.profile_nvtx is True                                                       # base_trainer.py:69 default
    tr = MultiviewTrainer(cfg, pipe, _Views(), device='cpu')
    tr.train_data_loader = _Loader()
    tr.validate = lambda: None
    tr.save_model = lambda: None
    del log[:]
    tr.train()
    assert emitted == [torch.cuda.is_available()]                                         # the profiler state needs a GPU runtime
    steps = log.count("MultiviewTrainer.step")
    assert steps >= 2 and log.count("MultiviewTrainer.backward") == steps - 1 and depth[0] == 0      # (the first call is the warm-up)
    i = log.index("MultiviewTrainer.backward")
    assert log[i - 1] in ("MultiviewTrainer.step", "Tracer.trace") or "MultiviewTrainer.step" in log[:i]
    # the fused step (modular tier on this CPU stand-in): same two names
    C_adamw = C.adamw_step_groups
    C.adamw_step_groups = stub._torch_adamw_groups
    try:
        ts = MultiviewTrainStep(stub._StubPipeline(), prune_every=-1)
        del log[:]
        ts.step(Rays(O, D), T)
    finally:
        C.adamw_step_groups = C_adamw
    assert log[0] == "MultiviewTrainer.step" and "MultiviewTrainer.backward" in log and depth[0] == 0
    # and the direct-issue tier brackets its backward launches likewise
    import inspect
    from wisp.trainers.multiview_trainer import _DirectNeRFStep
    body = inspect.getsource(_DirectNeRFStep.run)
    assert body.index('_range_push("MultiviewTrainer.backward")') < body.index("C.nerf_mlp_backward(") < body.index("_range_pop()")
