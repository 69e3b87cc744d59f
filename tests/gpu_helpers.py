"""Shared helpers of the GPU test files (tests/test_gpu_0_parity.py, tests/test_gpu_1_selfcheck.py): builders of small
pipelines with their CPU-oracle twins.  Test infrastructure only."""
import os

import numpy as np
import pytest
import torch

from oracle import hashgrid as ohash, nerf as onerf, raymarch as omarch, render as orender, spc as ospc


DEV = "cuda:0"


NGP_RES = [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512]


def _C():
    import wisp._C as C
    return C


def margin(name, value, limit):
    """assert value <= limit, and - with WISP_TEST_MARGINS=<file> - append (name, value, limit) to that file: the repetition
    harness (scripts/gpu_flaky.sh) collects how close every statistical threshold of the suite comes to tripping."""
    path = os.environ.get("WISP_TEST_MARGINS")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps({"name": name, "value": float(value), "limit": float(limit)}) + "\n")
    assert value <= limit, (name, value, limit)


def cuda(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
    t = t.to(DEV)
    return t if dtype is None else t.to(dtype)


def make_rays(n, seed, radius=3.0, spread=0.6):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = (radius * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    d = rng.uniform(-spread, spread, (n, 3)) - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def sparse_tree(level, n, seed):
    rng = np.random.default_rng(seed)
    oc = ospc.points_to_octree(rng.integers(0, 2 ** level, size=(n, 3)), level)
    pts, pyr, ex = ospc.octree_to_spc(oc)
    return oc, pts, pyr, ex


def _ray_like_coords(rng, n, dim=3, run=32):
    start = rng.uniform(-1, 1, (n // run, 1, dim))
    step = rng.normal(size=(n // run, 1, dim)) * 0.004
    return np.clip(start + step * np.arange(run)[None, :, None], -1, 1).reshape(n, dim).astype(np.float32)


# ------------------------------------------------------------------------------------------------ compositing
def _packs(lens, seed):
    rng = np.random.default_rng(seed)
    ridx = np.concatenate([np.full(n, r) for r, n in enumerate(lens) if n]).astype(np.int64)
    S = ridx.shape[0]
    return (ridx, rng.uniform(size=(S, 3)).astype(np.float32), rng.uniform(0, 40, size=(S, 1)).astype(np.float32),
            rng.uniform(1e-3, 0.05, size=(S, 1)).astype(np.float32), rng.uniform(1, 5, size=(S, 1)).astype(np.float32))


# ------------------------------------------------------------------------------------------------ end to end
def _build_pair(level=4, bitwidth=12, lods=16, hidden=64, dense=False):
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    torch.manual_seed(0)
    rng = np.random.default_rng(81)
    P = rng.integers(0, 2 ** level, size=(1500, 3))
    if dense:                                        # every cell of the level (nerf_hash.yaml:16-17 starts like this, at level 7)
        P = np.stack(np.meshgrid(*[np.arange(2 ** level)] * 3, indexing='ij'), -1).reshape(-1, 3)
    blas = OctreeAS.from_quantized_points(torch.from_numpy(P).short().to(DEV), level)
    grid = HashGrid.from_geometric(blas, feature_dim=2, num_lods=lods, multiscale_type='cat', feature_std=0.2,
                                   codebook_bitwidth=bitwidth, min_grid_res=4, max_grid_res=64)
    nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=hidden, num_layers=1, bias=True,
                              prune_density_decay=0.95, prune_min_density=0.5).to(DEV)
    onef = onerf.OracleNeRF(grid.resolutions, 2, bitwidth, 'cat', 0.2, hidden, 1, True, 4)
    sd = {k: v.detach().cpu() for k, v in nef.state_dict().items() if k in onef.state_dict()}
    onef.load_state_dict(sd, strict=False)
    oblas = onerf.OracleBLAS(ospc.points_to_octree(P, level))
    return nef, onef, oblas


def _dropin_trainer(pipe, amp, lr=1e-3, glw=500.0, loss='huber', rays_per_view=300):
    """wisp.trainers.MultiviewTrainer as app/nerf/main_nerf.py:110 builds it (nerf_hash.yaml's trainer block), over `pipe`."""
    from wisp.core import Rays
    from wisp.datasets import MultiviewTensorDataset, SampleRays
    from wisp.trainers import MultiviewTrainer, ConfigMultiviewTrainer, ConfigAdamW
    o, d = make_rays(rays_per_view, 5)
    ds = MultiviewTensorDataset(cuda(o)[None], cuda(d)[None], torch.rand(1, rays_per_view, 3, device=DEV), 1.0, 5.0,
                                transform=SampleRays(rays_per_view))
    cfg = ConfigMultiviewTrainer(optimizer=ConfigAdamW(lr=lr, eps=1e-16, weight_decay=1e-6), grid_lr_weight=glw, enable_amp=amp,
                                 prune_every=-1, rgb_loss_type=loss, rgb_loss_denom='rays', max_epochs=10, scheduler=False)
    return MultiviewTrainer(cfg, pipe, ds, device=DEV)


# ------------------------------------------------------------------------------------------------ fused decoder
def _decoder_pair(bias=True, in_dim=32, hidden=64):
    """A NeuralRadianceField whose decoders take `in_dim` grid features: the widths of the reference's app/nerf configs
    (32 = nerf_hash 'cat' 16x2, 12 = a 6-level 'cat' hash grid / the triplanar width, 5 = nerf_octree / nerf_codebook)."""
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import HashGrid, OctreeGrid
    from wisp.models.nefs import NeuralRadianceField
    torch.manual_seed(5)
    if in_dim == 5:
        grid = OctreeGrid(OctreeAS.make_dense(2), feature_dim=5, num_lods=2, multiscale_type='sum', feature_std=0.1)
    else:
        grid = HashGrid.from_geometric(OctreeAS.make_dense(2), feature_dim=2, num_lods=in_dim // 2, multiscale_type='cat',
                                       feature_std=0.1, codebook_bitwidth=10, min_grid_res=4, max_grid_res=64)
    nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=hidden, num_layers=1, bias=bias).to(DEV)
    assert nef.effective_feature_dim() == in_dim
    with torch.no_grad():
        for n, p in nef.named_parameters():
            if 'decoder' in n:
                p.mul_(2.0)            # wider activations so relu masks and the sigmoid are exercised
    return nef


def _check_fused_decoder(mode, io_dtype, tol, bias, in_dim, hidden, S):
    from wisp.ops.nerf_mlp import fused_nerf_decoder, supports
    nef = _decoder_pair(bias, in_dim, hidden)
    nef.decoder_compute = mode
    g = torch.Generator(device=DEV).manual_seed(1)
    feats = torch.randn(S, in_dim, device=DEV, generator=g)
    assert supports(nef, feats)
    dirs = torch.nn.functional.normalize(torch.randn(S, 3, device=DEV, generator=g), dim=1)
    w_rgb = torch.randn(S, 3, device=DEV, generator=g); w_den = torch.randn(S, 1, device=DEV, generator=g)

    f_ref = feats.clone().requires_grad_(True)
    dfeat = nef.decoder_density(f_ref)
    fdir = torch.cat([dfeat, nef.view_embedder(dirs)], dim=-1)
    rgb_ref = torch.sigmoid(nef.decoder_color(fdir[..., 1:])); den_ref = torch.relu(dfeat[..., 0:1])
    ((rgb_ref * w_rgb).sum() + (den_ref * w_den).sum()).backward()
    ref_grads = {n: p.grad.clone() for n, p in nef.named_parameters() if p.grad is not None}
    nef.zero_grad()
    # what plain torch bf16 autocast of the same modules loses against fp32 - the yardstick for the bf16 kernel
    f_amp = feats.clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        dfa = nef.decoder_density(f_amp)
        fda = torch.cat([dfa, nef.view_embedder(dirs)], dim=-1)
        rgb_a = torch.sigmoid(nef.decoder_color(fda[..., 1:])); den_a = torch.relu(dfa[..., 0:1])
    ((rgb_a.float() * w_rgb).sum() + (den_a.float() * w_den).sum()).backward()
    amp_err_feats = float((f_amp.grad - f_ref.grad).norm() / f_ref.grad.norm())
    amp_err = {n: float((p.grad - ref_grads[n]).norm() / ref_grads[n].norm()) for n, p in nef.named_parameters() if n in ref_grads}
    nef.zero_grad()

    f_in = feats.to(io_dtype).requires_grad_(True)
    rgb, den = fused_nerf_decoder(nef, f_in, dirs)
    assert rgb.dtype == torch.float32 and rgb.shape == (S, 3) and den.shape == (S, 1)
    ((rgb * w_rgb).sum() + (den * w_den).sum()).backward()
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), rgb_ref.detach().cpu().numpy(), atol=tol)       # fp32: 1e-4 contract
    np.testing.assert_allclose(den.detach().cpu().numpy(), den_ref.detach().cpu().numpy(), atol=tol * 10, rtol=tol)
    gs = float(f_ref.grad.abs().max())
    if mode == "fp32":
        assert float((f_in.grad.float() - f_ref.grad).abs().max()) <= 2e-4 * gs
    else:
        # bf16 activations flip a few relu masks near zero, which changes single gradient rows entirely: judge the
        # tensor by its relative L2 error and the bulk of its entries, not by the worst entry
        diff = (f_in.grad.float() - f_ref.grad)
        assert float(diff.norm() / f_ref.grad.norm()) <= 1.5 * amp_err_feats + 1e-2, (float(diff.norm() / f_ref.grad.norm()), amp_err_feats)
    for n, p in nef.named_parameters():
        if n in ref_grads:
            if mode == "fp32":
                scale = max(float(ref_grads[n].abs().max()), 1e-6)
                err = float((p.grad - ref_grads[n]).abs().max())
                assert err <= 3e-4 * scale, (n, err, scale)
            else:
                rel = float((p.grad - ref_grads[n]).norm() / ref_grads[n].norm())
                assert rel <= 1.5 * amp_err[n] + 1e-2, (n, rel, amp_err[n])
    assert f_in.grad.shape == (S, in_dim)
    r0, d0 = fused_nerf_decoder(nef, torch.zeros(0, in_dim, device=DEV), torch.zeros(0, 3, device=DEV))
    assert r0.shape == (0, 3) and d0.shape == (0, 1)


# ------------------------------------------------------------------------------------------------ octree / codebook grids
def _sparse_blas(level, n, seed):
    from wisp.accelstructs import OctreeAS
    rng = np.random.default_rng(seed)
    P = rng.integers(0, 2 ** level, size=(n, 3))
    blas = OctreeAS.from_quantized_points(torch.from_numpy(P).short().to(DEV), level)
    oblas = onerf.OracleBLAS(ospc.points_to_octree(P, level))
    return blas, oblas


def _assert_same_adam_trajectory(p1, p2, name, steps, max_lr):
    """Two runs of the same Adam steps whose gradients differ only by summation order.  With eps = 1e-15 an element whose
    gradient is rounding noise (|g| ~ 1e-9: e.g. a weight of a mostly inactive relu unit) still moves by the full +-lr, with
    the sign of the noise - so SOME elements may differ by up to steps x lr while the bulk agrees to rounding.  (The
    reference's torch.optim.Adam has the same property between two of its own runs.)  What is certain is the step bound;
    the outlier share is logged with a wide limit (gradients are compared directly, before the optimizer, by the callers)."""
    a, b = p1.detach().double().cpu(), p2.detach().double().cpu()
    diff = (a - b).abs()
    tol = 1e-4 * b.abs() + 2e-6
    assert float(diff.max()) <= 2.2 * steps * max_lr, name
    margin(f"adam trajectory outliers {name}", float((diff > tol).double().mean()), 0.4)
    margin(f"adam trajectory median {name}", float(diff.median()), 2e-6)


def snapshot_first_grad(trainer, store):
    """wrap trainer.optimizer_step so that the flat gradient buffer of every step is cloned into `store` before the update."""
    inner = trainer.optimizer_step

    def snap(*a, **kw):
        store.append(trainer.flat.grad.clone())
        return inner(*a, **kw)
    trainer.optimizer_step = snap
