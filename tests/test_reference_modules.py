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
