"""GPU (MI355X), collected AFTER tests/test_gpu_0_parity.py: the package against itself - fast paths against the modular paths
they replace, repeatability, scratch sizing, look-ahead issue, host plumbing.  None of these is an oracle comparison."""
import os

import numpy as np
import pytest
import torch

from oracle import hashgrid as ohash, nerf as onerf, raymarch as omarch, render as orender, spc as ospc
from gpu_helpers import *          # noqa: F401,F403  (DEV, cuda, make_rays, _build_pair, ...)
from gpu_helpers import margin, _C, _ray_like_coords, _packs, _build_pair, _dropin_trainer, _decoder_pair, _check_fused_decoder, _sparse_blas, \
    _assert_same_adam_trajectory

pytestmark = pytest.mark.gpu


def test_hashgrid_backward_slot_overflow_falls_back_to_atomics():
    """Adversarial batch for the binned backward: consecutive samples jump between 97 points of one small region, so there
    are no runs to merge and every tile sends its 8192 records per level to one or two buckets - far beyond the slot
    capacity.  The overflow path (memory-side atomics) must still give the oracle's gradient."""
    rng = np.random.default_rng(14)
    _, begin = ohash.table_layout(NGP_RES, 2 ** 19)
    shape = (int(begin[-1]), 2)
    pts = rng.uniform(0.30, 0.34, (97, 3)).astype(np.float32)
    n = 12288
    coords = pts[(np.arange(n) * 41) % 97]
    go = rng.normal(size=(n, 32)).astype(np.float32)
    want = ohash.hashgrid_backward(torch.from_numpy(coords), torch.from_numpy(go), shape, torch.from_numpy(begin), NGP_RES, 19,
                                   torch.float64)
    # fp32: the overflow path sums ~12 K fp32 atomics per entry in an order that changes from run to run (random-walk rounding
    # error ~ sqrt(N) * 2^-24 of the running sum: a few 1e-6 of the result's scale)
    for dt, tol in ((torch.float32, 1.5e-5), (torch.bfloat16, 1e-4)):
        g = torch.from_numpy(go).to(dt)
        want_d = want if dt == torch.float32 else ohash.hashgrid_backward(
            torch.from_numpy(coords), g.float(), shape, torch.from_numpy(begin), NGP_RES, 19, torch.float64)
        got = _C().hashgrid_interpolate_backward(cuda(coords), g.to(DEV), shape, cuda(begin), NGP_RES, 19)
        scale = float(want_d.abs().max())
        assert float((got.double().cpu() - want_d).abs().max()) <= tol * scale, dt


@pytest.mark.parametrize("res,bitwidth", [([16, 64, 300, 1024, 2048, 8192], 19), (NGP_RES, 19)])
@pytest.mark.parametrize("n", [4096, 8192, 65536])
def test_hashgrid_backward_is_repeatable_and_race_free(res, bitwidth, n):
    """Round-1 failure (GPUTEST_r01: error 1.7 on a gradient of scale 11.8): buckets of a dense level sized (res+1)^3 reached
    into the next level's rows and its zero-adding flush raced with the owner of those rows - a LOST UPDATE, which showed up
    only on some boxes.  50 repetitions of the same backward, each one compared with the float64 oracle at float
    add-order tolerance (a lost update is five orders of magnitude above it).  Repetitions are not required to be bitwise
    equal: a slot that overflows falls back to float atomics, whose order is free."""
    rng = np.random.default_rng(5 + n)
    _, begin = ohash.table_layout(res, 2 ** bitwidth)
    shape = (int(begin[-1]), 2)
    coords = _ray_like_coords(rng, n)
    go = rng.normal(size=(n, len(res) * 2)).astype(np.float32)
    c, g, b = cuda(coords), cuda(go), cuda(begin)
    for dt, tol, reps in ((torch.float32, 4e-6, 50), (torch.bfloat16, 3e-5, 25)):
        gd = g.to(dt)
        want = ohash.hashgrid_backward(torch.from_numpy(coords), gd.float().cpu(), shape, torch.from_numpy(begin), res, bitwidth,
                                       torch.float64)
        scale = float(want.abs().max())
        want = want.to(DEV)
        worst = 0.0
        for rep in range(reps):
            got = _C().hashgrid_interpolate_backward(c, gd, shape, b, res, bitwidth)
            err = float((got.double() - want).abs().max())
            worst = max(worst, err)
            assert err <= tol * scale, f"{dt} repetition {rep}: |grad - oracle| = {err} on a gradient of scale {scale}"
        print(f"n={n} {dt}: worst |grad - oracle| over {reps} repetitions = {worst:.3e} (scale {scale:.3f})")


def test_hashgrid_backward_scratch_follows_what_the_launches_fill(monkeypatch):
    """the record slots of the binned backward start at the no-merge expectation (3 GB of scratch for 0.45 GB of
    records at 2 M ray-ordered samples) and are then sized from the fullest slot the launches really produced
    (wisp._C._SlotFit over wisp_hashgrid_bwd_slot_stats).  After the fit: a fraction of the scratch, the gradient unchanged,
    no slot overflow; and a fit that is far too small (forced) costs nothing but speed - same gradient through the atomic
    path - and grows back."""
    import ctypes
    import synlego
    from wisp.accelstructs import OctreeAS
    from wisp.core import Rays
    C = _C()
    monkeypatch.setattr(C._SlotFit, "CHECK_EVERY", 1)
    monkeypatch.setattr(C._SlotFit, "ADOPT_AFTER", 0)         # adopt a check at the very next launch (production: 8 launches later)
    C._slot_fits.clear()
    cells = synlego.occupied_cells(7, device=DEV)
    blas = OctreeAS.from_quantized_points(cells, 7)
    o, d, _ = synlego.ray_bank(16384, seed=5, device=DEV, with_gt=False)
    rm = blas.raymarch(Rays(o, d, dist_min=1.0, dist_max=5.0), 'ray', 2048)
    coords = rm.samples
    n = coords.shape[0]
    assert n > 500_000
    _, begin = ohash.table_layout(NGP_RES, 2 ** 19)
    shape = (int(begin[-1]), 2)
    g = (torch.randn(n, 32, device=DEV) * 1e-3).bfloat16()
    b = cuda(begin)
    run = lambda: C.hashgrid_interpolate_backward(coords, g, shape, b, NGP_RES, 19, zero_from_col=30)
    first = run()
    torch.cuda.synchronize()
    fit = next(iter(C._slot_fits.values()))
    arr = (ctypes.c_int32 * 16)(*NGP_RES)
    full = int(C.lib.wisp_hashgrid_bwd_workspace_bytes(n, 3, C.BF16, 2, arr, 16, 19, None))
    for _ in range(3):
        got = run()
        torch.cuda.synchronize()
    st = fit.last
    assert st is not None and all(f < c for f, c, bse in zip(st["fill"], st["cap"], st["base"]) if bse > 0), st   # no overflow after the fit
    scales = (ctypes.c_float * 16)(*fit.scale)
    fitted = int(C.lib.wisp_hashgrid_bwd_workspace_bytes(n, 3, C.BF16, 2, arr, 16, 19, ctypes.cast(scales, ctypes.c_void_p)))
    ws = C._bwd_ws[(torch.device(DEV), C._stream().value)]
    written = 8 * sum(st["records"])                          # compact records: 8 bytes each
    # Capacity follows the FULLEST slot of a level (x 1.2), the bytes written are the sum over all slots.  On dense levels a bucket
    # of consecutive rows is a slab of space and the slabs a workgroup's rays cross receive several times the average: those
    # levels deal strips of 32 rows to their buckets instead (BinLevels::strip_magic, round 5), which brought this shape from
    # 5.5 x the records to 4 x (the bench's 2^21-sample step, with fuller slots and the wide emitter: 4.6 x -> 2.5 x).  What remains is the clumped-
    # Poisson spread of 100-250 records per slot on the hashed levels (fullest 1.9-2.6 x the mean).
    assert written > (100 << 20) and st["workspace_bytes"] <= 5 * written, (st["workspace_bytes"], written)
    assert fitted <= 5 * written and fitted < 0.7 * full and ws.numel() <= 3 * fitted + (64 << 20), (fitted, full, ws.numel())
    ref = first.double()
    assert float((got.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())          # same gradient before and after the fit
    print(f"scratch: unscaled {full / 2**30:.2f} GiB -> fitted {fitted / 2**30:.2f} GiB for {written / 2**30:.2f} GiB of records; "
          f"scales {[round(x, 3) for x in fit.scale]}")
    # forced far too small: every slot overflows into atomics - same numbers - and the next check grows the slots again
    fit.pending = None                                        # (drop the check of the last launch: it would replace the forced sizes)
    fit.scale = [0.02] * 16
    small = run()
    torch.cuda.synchronize()
    assert float((small.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    run(); torch.cuda.synchronize(); run(); torch.cuda.synchronize()
    assert max(fit.scale) > 0.03


def test_query_chain_equals_the_query_columns_with_any_hint():
    """wisp_spc_query_chain: columns first_level .. level of the parents query (the oracle's), with no hints, with the right cell of
    first_level as hint, with wrong cells, with -1 and with hints for groups of consecutive coordinates - a hint may shorten the
    walk, never change a result (points outside, on cell faces and NaN included)."""
    oc, pts, pyr, ex = sparse_tree(6, 3000, 21)
    rng = np.random.default_rng(23)
    n = 60000
    x = rng.uniform(-1.05, 1.05, (n, 3)).astype(np.float32)
    x[:6] = [[1, 1, 1], [-1, -1, -1], [0, 0, 0], [np.nan, 0, 0], [1.0000001, 0, 0], [-0.0, 0.5, -0.5]]
    x[6:1006] = (rng.integers(-64, 65, (1000, 3)) / 64.0).astype(np.float32)
    # half of the coordinates inside occupied cells of level 6 (so that the deep columns are not all -1)
    leaf = pts[pyr[1, 6]:pyr[1, 6] + pyr[0, 6]].astype(np.float32)
    pick = leaf[rng.integers(0, leaf.shape[0], n // 2)]
    x[n // 2:] = ((pick + rng.uniform(0.0, 1.0, pick.shape)) / 32.0 - 1.0).astype(np.float32)
    C = _C()
    for level, first in ((6, 3), (6, 6), (6, 0), (5, 2)):
        want = ospc.query(oc, ex, x, level, with_parents=True)[:, first:]
        got = C.spc_query_chain(cuda(oc), cuda(ex), cuda(pts), cuda(x), level, first)
        assert got.dtype == torch.int64 and tuple(got.shape) == (n, level - first + 1) and np.array_equal(got.cpu().numpy(), want)
        right = want[:, 0].astype(np.int32)
        wrong = right.copy()
        lo, cnt = int(pyr[1, first]), int(pyr[0, first])
        flip = rng.uniform(size=n) < 0.3
        wrong[flip] = rng.integers(lo, lo + cnt, int(flip.sum())).astype(np.int32)           # some other cell of that level
        wrong[rng.uniform(size=n) < 0.1] = -1
        for hint in (right, wrong, np.full(n, -1, np.int32)):
            got = C.spc_query_chain(cuda(oc), cuda(ex), cuda(pts), cuda(x), level, first, hint=cuda(hint), hint_group=1)
            assert np.array_equal(got.cpu().numpy(), want)
        for group in (4, 16):                                 # one hint per run of `group` coordinates: right for some of them only
            hint = right[::group].copy()
            got = C.spc_query_chain(cuda(oc), cuda(ex), cuda(pts), cuda(x), level, first, hint=cuda(hint), hint_group=group)
            assert np.array_equal(got.cpu().numpy(), want)
    assert tuple(C.spc_query_chain(cuda(oc), cuda(ex), cuda(pts), cuda(x[:0]), 6, 3).shape) == (0, 4)


@pytest.mark.parametrize("cap", [0, 3, 64])
def test_raytrace_nugget_cache_and_level_extremes(cap, monkeypatch):
    """The count phase parks `cap` nuggets per ray and the emit phase copies them; rays with more are walked again, cap 0
    walks twice - every combination must give the oracle's nuggets.  Levels 0 and 1 (root only / root's children) and a
    level-6 dense tree (up to ~190 nuggets per ray, far beyond any cap) are the extremes of the group traversal."""
    C = _C()
    monkeypatch.setattr(C, "RAYTRACE_CACHE_CAP", cap)
    cases = [(sparse_tree(6, 30000, 35), 6), (sparse_tree(1, 5, 36), 1),
             ((ospc.create_dense_octree(6),) + ospc.octree_to_spc(ospc.create_dense_octree(6)), 6)]
    for (oc, pts, pyr, ex), level in cases:
        o, d = make_rays(1500, 37 + level)
        want = ospc.raytrace(oc, pts, pyr, ex, o, d, level, with_exit=True)
        ridx, pidx, depth, offsets = C.spc_raytrace(cuda(oc), cuda(pts), cuda(ex), cuda(o), cuda(d), level, True)
        assert np.array_equal(ridx.cpu().numpy(), want[0]) and np.array_equal(pidx.cpu().numpy(), want[1])
        assert np.array_equal(depth.cpu().numpy(), want[2])
        # level 0: the root cell alone
        w0 = ospc.raytrace(oc, pts, pyr, ex, o, d, 0, with_exit=False)
        r0, p0, d0, _ = C.spc_raytrace(cuda(oc), cuda(pts), cuda(ex), cuda(o), cuda(d), 0, False)
        assert np.array_equal(r0.cpu().numpy(), w0[0]) and np.array_equal(d0.cpu().numpy(), w0[2]) and int(p0.abs().max()) == 0


def test_raymarch_ray_coarse_pretest_changes_nothing_at_flagship_shape():
    """nerf_hash.yaml shape (level 7, 2048 candidates, near/far 1/5, in-kernel jitter): with and without the coarse level."""
    from wisp.accelstructs import OctreeAS
    from wisp.core import Rays
    rng = np.random.default_rng(45)
    centres = rng.uniform(20, 108, size=(12, 1, 3))
    cells = np.clip(centres + rng.normal(0, 5.0, size=(12, 8000, 3)), 0, 127).reshape(-1, 3).astype(np.int64)
    oc = ospc.points_to_octree(cells, 7)
    blas = OctreeAS(cuda(oc))
    assert blas.pyramid[0, 4] < 0.5 * 4096          # most level-4 cells are empty: the pre-test has something to skip
    o, d = make_rays(3000, 46)
    rays = Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0)
    coarse, lc = blas._coarse_bitfield(rays, 2048, 7)
    assert lc == 4 and coarse is not None
    args = (blas._bitfield(7), blas.octree, blas.prefix, rays.origins, rays.dirs, 1.0, 5.0, 2048, 7, None, 123)
    a = _C().raymarch_ray(*args)
    b = _C().raymarch_ray(*args, coarse_bits=coarse, coarse_level=lc)
    assert a[0].shape[0] > 10000
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("kind", ["huber", "l2", "l1"])
def test_composite_loss_in_one_launch_equals_the_three_launches(kind):
    """wisp_composite_loss (compositing + photometric loss + compositing backward of a training step, one pass per ray) against
    wisp_composite_fwd -> wisp_rgb_loss -> wisp_composite_bwd on the same packed samples: the composited colours and both
    gradients bit for bit (same arithmetic in the same order per ray), the loss to rounding (its terms are grouped per ray);
    rays without samples, with 1, 63, 64, 65 and several hundred samples, errors on both sides of the huber knee; and the value
    is reproducible from call to call (ticket reduction in index order)."""
    C = _C()
    rng = np.random.default_rng(907)
    lens = [0, 1, 63, 64, 65, 130, 700, 0, 5] + list(rng.integers(0, 90, 3000))
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    S, R = int(offs[-1]), len(lens)
    color = rng.uniform(0, 1, (S, 3)).astype(np.float32)
    dens = (rng.uniform(0, 30, (S, 1)) * (rng.uniform(size=(S, 1)) < 0.6)).astype(np.float32)
    delt = rng.uniform(1e-3, 4e-2, (S, 1)).astype(np.float32)
    gts = rng.uniform(-1.5, 2.5, (R, 3)).astype(np.float32)          # some |rgb - gt| > 1: the linear part of huber
    bg = (0.2, 0.5, 0.9)
    c, d, dl, o, g = cuda(color), cuda(dens), cuda(delt), cuda(offs), cuda(gts)
    rgb, _a, _d, _h, _w = C.composite_fwd(c, d, dl, None, None, o, R, bg)
    loss3, g_rgb = C.rgb_loss(rgb, g, kind)
    gc3, gd3 = C.composite_bwd(g_rgb, None, None, c, d, dl, None, None, o, bg)
    loss1, gc1, gd1, rgb1 = C.composite_loss(c, d, dl, o, R, bg, g, kind, with_rgb=True)
    # (the density gradient to the last bit or two: the compiler contracts `G T e - suffix` into fused multiply-adds its own way
    #  in each kernel)
    assert torch.equal(rgb1, rgb) and torch.equal(gc1, gc3)
    torch.testing.assert_close(gd1, gd3, rtol=2e-6, atol=1e-6 * float(gd3.abs().max()))
    assert float(gd3.abs().max()) > 0 and abs(float(loss1) - float(loss3)) <= 2e-6 * abs(float(loss3))
    for _ in range(3):
        again = C.composite_loss(c, d, dl, o, R, bg, g, kind)
        assert float(again[0]) == float(loss1) and torch.equal(again[1], gc1) and torch.equal(again[2], gd1) and again[3] is None
    # more rays than workgroups (grid-stride over rays) and a single ray
    big = np.concatenate([[0], np.cumsum(rng.integers(0, 6, 20000))]).astype(np.int64)
    Sb = int(big[-1])
    cb, db, dlb = cuda(rng.uniform(0, 1, (Sb, 3)).astype(np.float32)), cuda(rng.uniform(0, 20, (Sb, 1)).astype(np.float32)), cuda(np.full((Sb, 1), 0.01, np.float32))
    gb = cuda(rng.uniform(0, 1, (20000, 3)).astype(np.float32))
    rgbb = C.composite_fwd(cb, db, dlb, None, None, cuda(big), 20000, bg)[0]
    l3, grb = C.rgb_loss(rgbb, gb, kind)
    want = C.composite_bwd(grb, None, None, cb, db, dlb, None, None, cuda(big), bg)
    got = C.composite_loss(cb, db, dlb, cuda(big), 20000, bg, gb, kind)
    assert torch.equal(got[1], want[0]) and abs(float(got[0]) - float(l3)) <= 2e-6 * abs(float(l3))
    torch.testing.assert_close(got[2], want[1], rtol=2e-6, atol=1e-6 * float(want[1].abs().max()))
    one = C.composite_loss(c[:700], d[:700], dl[:700], cuda(np.int64([0, 700])), 1, bg, g[:1], kind, with_rgb=True)
    assert np.isfinite(float(one[0])) and tuple(one[3].shape) == (1, 3)
    # and against the oracle itself: oracle.render.composite in float64 + the trainer's loss (multiview_trainer.py:140-154), autograd
    ridx = np.repeat(np.arange(R), lens).astype(np.int64)
    c64 = torch.from_numpy(color).double().requires_grad_(True)
    d64 = torch.from_numpy(dens).double().requires_grad_(True)
    b = torch.from_numpy(ospc.mark_pack_boundaries(ridx))
    want = orender.composite(c64, d64, torch.from_numpy(delt), torch.from_numpy(delt), torch.from_numpy(ridx), b, R, bg, with_depth=False)
    x = want["rgb"] - torch.from_numpy(gts).double()
    if kind == "huber":
        per = torch.where(x.abs() < 1.0, 0.5 * x * x, x.abs() - 0.5)
    else:
        per = x * x if kind == "l2" else x.abs()
    ref_loss = per.mean()
    ref_loss.backward()
    assert abs(float(loss1) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
    np.testing.assert_allclose(rgb1.cpu().numpy(), want["rgb"].detach().numpy(), rtol=0, atol=1e-5)     # contract: 1e-4
    gc_ref, gd_ref = c64.grad.numpy(), d64.grad.numpy()
    np.testing.assert_allclose(gc1.cpu().numpy(), gc_ref, rtol=0, atol=1e-5 * np.abs(gc_ref).max())
    np.testing.assert_allclose(gd1.cpu().numpy(), gd_ref, rtol=0, atol=2e-5 * np.abs(gd_ref).max())     # (as test_composite_forward_backward)


def test_adamw_groups_equal_one_launch_per_group():
    """wisp_adamw_step_groups (all optimizer param groups of the flat buffer in one launch) against wisp_adamw_step run once
    per group: bit-identical parameters, moments, zeroed gradients and bf16 shadow; group lengths not multiples of 4."""
    torch.manual_seed(4)
    lens = [10259, 70001, 6]                      # decoder-like, grid-like, a tiny 'rest'
    begins, off = [], 0
    for k in lens:
        begins.append(off); off += (k + 3) // 4 * 4
    n = off
    base = [torch.randn(n, device=DEV) for _ in range(2)]
    lrs, wds = [1e-2, 5.0, 3e-3], [1e-3, 0.0, 1e-2]
    def fresh():
        return base[0].clone(), base[1].clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pa, ga, ma, va = fresh()
    pb, gb, mb, vb = fresh()
    sh_a = torch.zeros(lens[1], dtype=torch.bfloat16, device=DEV); sh_b = torch.zeros_like(sh_a)
    for step in (1, 2):
        ga.copy_(base[1] * step); gb.copy_(base[1] * step)
        for k in range(3):
            a, b = begins[k], begins[k] + lens[k]
            _C().adamw_step(pa[a:b], ga[a:b], ma[a:b], va[a:b], lrs[k], 0.9, 0.99, 1e-15, wds[k], step, grad_scale=0.25,
                            zero_grad=True, bf16_shadow=sh_a if k == 1 else None)
        _C().adamw_step_groups(pb, gb, mb, vb, [(begins[k], lens[k], lrs[k], wds[k], sh_b if k == 1 else None) for k in range(3)],
                               0.9, 0.99, 1e-15, step, grad_scale=0.25, zero_grad=True)
        for x, y in ((pa, pb), (ma, mb), (va, vb), (sh_a, sh_b)):
            assert torch.equal(x, y)
        for k in range(3):
            assert float(gb[begins[k]:begins[k] + lens[k]].abs().max()) == 0.0
        pad = begins[1] - lens[0]                                   # padding between groups is never touched
        assert pad > 0 and torch.equal(pb[lens[0]:begins[1]], base[0][lens[0]:begins[1]])


def test_adamw_groups_off_a_16_byte_boundary_and_more_than_four_of_them():
    """Groups may begin anywhere (the rows the hash-grid backward's folded update leaves over start where a table level starts)
    and there may be more than one launch's worth of them: bit-identical to one wisp_adamw_step per group."""
    torch.manual_seed(5)
    n = 40000
    spans = [(0, 1001), (1002, 3000), (4003, 9), (4013, 20001), (24014, 7), (24022, 15978)]     # begins 0, 2, 3, 1, 2, 2 mod 4
    base = [torch.randn(n, device=DEV) for _ in range(2)]
    pa, ma, va = base[0].clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb, mb, vb = base[0].clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sh_a = torch.zeros(n, dtype=torch.bfloat16, device=DEV); sh_b = torch.zeros_like(sh_a)
    for step in (1, 2, 3):
        ga, gb = (base[1] * step).clone(), (base[1] * step).clone()
        for a, k in spans:
            # (wisp_adamw_step wants 16-byte aligned buffers: run it on an aligned copy of the span)
            tp, tg, tm, tv = pa[a:a + k].clone(), ga[a:a + k].clone(), ma[a:a + k].clone(), va[a:a + k].clone()
            tsh = torch.zeros(k, dtype=torch.bfloat16, device=DEV)
            _C().adamw_step(tp, tg, tm, tv, 3e-2, 0.9, 0.99, 1e-15, 1e-2, step, grad_scale=0.5, zero_grad=True, bf16_shadow=tsh)
            pa[a:a + k], ga[a:a + k], ma[a:a + k], va[a:a + k], sh_a[a:a + k] = tp, tg, tm, tv, tsh
        _C().adamw_step_groups(pb, gb, mb, vb, [(a, k, 3e-2, 1e-2, sh_b[a:a + k]) for a, k in spans], 0.9, 0.99, 1e-15, step,
                               grad_scale=0.5, zero_grad=True)
        for x, y in ((pa, pb), (ma, mb), (va, vb), (sh_a, sh_b), (ga, gb)):
            assert torch.equal(x, y)
    gaps = torch.ones(n, dtype=torch.bool, device=DEV)
    for a, k in spans:
        gaps[a:a + k] = False
    assert int(gaps.sum()) > 0 and torch.equal(pb[gaps], base[0][gaps])         # nothing outside the groups is touched


@pytest.mark.parametrize("amp", [True, False])
def test_grid_optimizer_folded_into_the_backward_equals_the_separate_pass_bit_for_bit(amp):
    """MultiviewTrainStep on one GPU folds the table's AdamW step into the hash-grid backward's reduce kernel
    (_fused_update_args).  Same model, same rays, six steps incl. a learning-rate milestone, with and without: parameters, both
    moments and the bf16 copy are bit-identical after every step wherever only the fixed-point bucket sums feed the update, the
    gradient buffer ends zeroed, and the folded path really ran."""
    import copy
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    torch.manual_seed(0)
    cells = np.random.default_rng(81).integers(0, 32, size=(5000, 3))
    blas = OctreeAS.from_quantized_points(torch.from_numpy(cells).short().to(DEV), 5)
    # nerf_hash.yaml's table: levels 7..14 have more than 32 buckets, i.e. one reduce workgroup per bucket
    grid = HashGrid.from_geometric(blas, feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=0.1, codebook_bitwidth=19,
                                   min_grid_res=16, max_grid_res=512)
    nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1, bias=True,
                              prune_density_decay=0.95, prune_min_density=0.5).to(DEV)
    nef2 = copy.deepcopy(nef)
    R, NS = 700, 1024               # dense sampling: consecutive samples share fine cells, as in training (see below)
    o, d = make_rays(R, 711)
    gts = cuda(np.random.default_rng(713).uniform(size=(R, 3)).astype(np.float32))
    rays = Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0)
    kw = dict(prune_every=-1, enable_amp=amp, scheduler_milestones=[3], scheduler_gamma=0.5)
    tr1 = MultiviewTrainStep(Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=NS, bg_color=(0, 0, 0))), **kw)
    tr2 = MultiviewTrainStep(Pipeline(nef2, PackedRFTracer(raymarch_type='ray', num_steps=NS, bg_color=(0, 0, 0))), **kw)
    assert tr1._direct is not None and tr1._direct.hash_fast and tr1.fuse_grid_optimizer
    tr2.fuse_grid_optimizer = False
    # record slots at their unscaled size: a slot fit learned by an earlier test of this table shape (another scene, another
    # sampling density) would make many more slots overflow into the order-free atomic path than this scene does on its own
    _C()._slot_fits.clear()
    left_over = []
    inner = tr1._uncovered_grid_ranges
    tr1._uncovered_grid_ranges = lambda cover: left_over.append(inner(cover)) or left_over[-1]
    ga, gb = tr1.flat.ranges["grid"]
    for k in range(6):
        jit = cuda(np.random.default_rng(720 + k).uniform(size=(R, NS)).astype(np.float32))
        l1, s1 = tr1.step(rays, gts, jitter=jit)
        l2, s2 = tr2.step(rays, gts, jitter=jit)
        assert s1 == s2 >= 4096 and float(l1) == float(l2)           # same state going in: same forward
        assert tr1.fused_elements_last > 0 and tr2.fused_elements_last == 0 and len(left_over) == k + 1
        folded = torch.zeros(tr1.flat.data.numel(), dtype=torch.bool, device=DEV)
        folded[ga:gb] = True
        for lo, hi in left_over[-1]:
            folded[lo:hi] = False
        assert int(folded.sum()) == tr1.fused_elements_last >= 0.5 * (gb - ga)     # every level with one owner per bucket
        # Bit-exact where nothing but the fixed-point bucket sums feeds the update: the hashed levels.  (A record slot that
        # overflows sends its excess through float atomics, whose order is free - in both trainers.  The one dense level with
        # single-owner buckets - 80^3 rows, a bucket = a slab of space - does that on this clustered little scene; on the
        # hashed ones a slot receives about its no-merge expectation +- 9 % here, so a few of the 9 K slots spill a record or two.)
        first = tr1._direct._first_idx_host
        off = next(o for p, o in tr1.flat._grid_params if p is tr1._direct.table)
        exact = folded.clone()
        exact[:off + first[8] * 2] = False
        assert int(exact.sum()) == 8 * (1 << 19) * 2             # levels 8..14 + the gradient-free finest level (tail workgroups of the launch)
        for name in ("exp_avg", "exp_avg_sq", "data"):
            x, y = getattr(tr1.flat, name)[exact], getattr(tr2.flat, name)[exact]
            differ = x != y
            # the same bits - except at the handful of entries (measured: 30-120 of 7.3 M) that received a record through the
            # float-atomic fall-back of a full slot; there the two runs are two orders of the same sum
            margin(f"folded adamw, step {k}: hashed-level entries of {name} not bit-identical to the separate pass",
                   float(differ.float().mean()), 2e-4)
            if bool(differ.any()):
                scale = float(y.abs().max())
                lim = 2.5 * tr1.lr * tr1.grid_lr_weight if name == "data" else 1e-5 * scale
                assert float((x[differ] - y[differ]).abs().max()) <= lim, (k, name)
        for name in ("data", "exp_avg", "exp_avg_sq"):
            # everywhere else both trainers ran the same separate pass on gradients that differ by float-atomic order at most
            # (the coarse levels' buckets are flushed by several workgroups)
            x, y = getattr(tr1.flat, name), getattr(tr2.flat, name)
            tol = 1e-5 * float(y.abs().max()) if name != "data" else 2.5 * tr1.lr * tr1.grid_lr_weight
            assert float((x[~exact] - y[~exact]).abs().max()) <= tol, (k, name)
        assert float(tr1.flat.grad.abs().max()) == 0.0 and float(tr2.flat.grad.abs().max()) == 0.0
        if amp:
            for tr in (tr1, tr2):                        # the bf16 copy is the rounded master, whoever wrote it
                assert torch.equal(tr.flat.shadow, tr.flat.data[ga:gb].bfloat16())
        # next step from the same state again (the bits the atomic order left different would otherwise spread)
        for name in ("data", "exp_avg", "exp_avg_sq") + (("shadow",) if amp else ()):
            getattr(tr2.flat, name).copy_(getattr(tr1.flat, name))
    assert tr1.opt_steps == tr2.opt_steps == 6


def test_dropin_regime_overflowing_loss_scale_skips_the_step_and_backs_off():
    """found_inf handling end to end: with a loss scale of 2^40 the fp16 gradients entering the decoder / hash-grid backward
    overflow; the table gradient must come out non-finite (not wrapped into a finite number by the fixed-point bins),
    GradScaler.step must skip the optimizer (parameters bit-identical) and update() must halve the scale."""
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.tracers import PackedRFTracer
    nef, _, _ = _build_pair()
    R, steps = 300, 96
    o, d = make_rays(R, 93)
    gts = np.random.default_rng(94).uniform(size=(R, 3)).astype(np.float32)
    pipe = Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=steps, bg_color=(0.0, 0.0, 0.0)))
    tr = _dropin_trainer(pipe, amp=True)
    tr.scaler = torch.amp.GradScaler('cuda', init_scale=2.0 ** 40)
    data = {"rays": Rays(cuda(o)[None], cuda(d)[None], dist_min=1.0, dist_max=5.0), "rgb": cuda(gts)[None]}
    with torch.autocast('cuda', enabled=True):
        tr.step(data)
        before = {n: p.detach().clone() for n, p in nef.named_parameters()}
        tr.step(data)
    table_grad = nef.grid.codebook.feats.grad
    assert table_grad is not None and not bool(torch.isfinite(table_grad).all()), "overflow was laundered into finite numbers"
    for n, p in nef.named_parameters():
        assert torch.equal(p.detach(), before[n]), f"{n} moved although the step had to be skipped"
    assert tr.scaler.get_scale() == 2.0 ** 39
    # and the trainer recovers: after enough back-offs a step goes through
    with torch.autocast('cuda', enabled=True):
        for _ in range(40):
            tr.step(data)
            if any(not torch.equal(p.detach(), before[n]) for n, p in nef.named_parameters()):
                break
    assert any(not torch.equal(p.detach(), before[n]) for n, p in nef.named_parameters())
    assert tr.scaler.get_scale() < 2.0 ** 39


@pytest.mark.parametrize("kind,march,steps", [("hash", "voxel", 6), ("octree", "voxel", 6), ("codebook", "voxel", 6), ("codebook", "ray", 96)])
def test_dropin_regime_other_configs_fp16_autocast_tracks_fp32(kind, march, steps):
    """The unchanged trainer over the other BASELINE configurations' pieces (C4: hash grid + 'voxel' march; nerf_octree /
    C5 VQAD: OctreeGrid / CodebookOctreeGrid, bias-free decoders, RMSprop): three iterations of wisp.trainers.MultiviewTrainer with
    enable_amp (fp16 autocast + GradScaler) next to the same trainer without amp from the same state and the same in-kernel
    jitter seeds: same sample counts, losses within fp16 / bf16-decoder tolerance, no overflow (the scale stands), finite
    parameters - the kernels under the octree and codebook grids compute in fp32 whatever the ambient autocast dtype."""
    import copy
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.models.grids import CodebookOctreeGrid, OctreeGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    from wisp.datasets import MultiviewTensorDataset, SampleRays
    from wisp.trainers import MultiviewTrainer, ConfigMultiviewTrainer, ConfigAdamW, ConfigRMSprop
    if kind == "hash":
        nef, _, _ = _build_pair(lods=16)
    else:
        blas, _ = _sparse_blas(5, 3000, 131)
        torch.manual_seed(3)
        if kind == "octree":
            grid = OctreeGrid(blas, feature_dim=5, num_lods=4, multiscale_type='sum', feature_std=0.5)
        else:
            grid = CodebookOctreeGrid(blas, feature_dim=5, num_lods=4, multiscale_type='sum', feature_std=0.7, codebook_bitwidth=4)
        nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1,
                                  bias=(kind != "codebook")).to(DEV)
    R = 400
    o, d = make_rays(R, 391)
    gts = cuda(np.random.default_rng(393).uniform(size=(R, 3)).astype(np.float32))
    data = {"rays": Rays(cuda(o)[None], cuda(d)[None], dist_min=1.0, dist_max=5.0), "rgb": gts[None]}
    runs = {}
    for amp in (False, True):
        pipe = Pipeline(copy.deepcopy(nef), PackedRFTracer(raymarch_type=march, num_steps=steps, bg_color=(1.0, 1.0, 1.0)))
        ds = MultiviewTensorDataset(cuda(o)[None], cuda(d)[None], gts[None], 1.0, 5.0, transform=SampleRays(R))
        oc = ConfigRMSprop(lr=1e-3, eps=1e-8) if kind == "codebook" else ConfigAdamW(lr=1e-3, eps=1e-16, weight_decay=1e-6)
        cfg = ConfigMultiviewTrainer(optimizer=oc, grid_lr_weight=100.0, enable_amp=amp, prune_every=-1,
                                     rgb_loss_type='l2' if kind == "codebook" else 'huber', max_epochs=10)
        tr = MultiviewTrainer(cfg, pipe, ds, device=DEV)
        losses, counts = [], []
        with torch.autocast('cuda', enabled=amp):
            tr.step(data)                                            # warm-up call
            for k in range(3):
                torch.manual_seed(50 + k)                            # the marches draw their jitter seed from torch's generator
                before = tr.tracker.metrics.rgb_loss
                tr.step(data)
                losses.append(tr.tracker.metrics.rgb_loss - before)
                counts.append(pipe.tracer.get_prev_num_samples())
        assert all(torch.isfinite(p).all() for p in pipe.nef.parameters())
        if amp:
            assert tr.scaler.get_scale() == 65536.0
        runs[amp] = (losses, counts)
    assert runs[True][1] == runs[False][1] and min(runs[True][1]) > 1000
    np.testing.assert_allclose(runs[True][0], runs[False][0], rtol=3e-2)
    assert runs[False][0][2] < runs[False][0][0]                       # and it trains


@pytest.mark.parametrize("io_dtype,in_dim", [(torch.bfloat16, 32), (torch.float16, 32), (torch.float32, 32), (torch.float32, 5),
                                             (torch.bfloat16, 12)])
def test_decoder_with_per_ray_view_code_equals_per_sample_directions(io_dtype, in_dim):
    """wisp_nerf_mlp_{fwd,bwd}_rays: the view direction encoded once per ray (wisp_nerf_mlp_dir_code) and gathered by ray index
    inside the kernels must give exactly what the per-sample entry points give on directions gathered like
    packed_rf_tracer.py:70-76 does - same arithmetic, so bit-identical outputs and gradients.  Every row shape the per-sample
    kernels take: the 32-wide 16-bit rows of nerf_hash, fp32 rows, and the narrow rows of the octree / codebook / triplanar
    fields (5 and 12 features)."""
    C = _C()
    rng = np.random.default_rng(91)
    R, S = 4097, 200003
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ray_dirs = cuda(d)
    ridx = torch.from_numpy(np.sort(rng.integers(0, R, S))).to(DEV)
    feats = torch.from_numpy(rng.normal(size=(S, in_dim)).astype(np.float32) * 0.5).to(DEV).to(io_dtype)
    n = int(C.lib.wisp_nerf_mlp_param_count(in_dim, 64, 4))
    params = torch.from_numpy(rng.normal(size=n).astype(np.float32) * 0.2).to(DEV)
    g_rgb = torch.from_numpy(rng.normal(size=(S, 3)).astype(np.float32)).to(DEV)
    g_den = torch.from_numpy(rng.normal(size=(S, 1)).astype(np.float32)).to(DEV)
    assert C.nerf_mlp_rays_supported(io_dtype, in_dim, 64, 4, True)
    assert not C.nerf_mlp_rays_supported(io_dtype, in_dim, 64, 4, False)          # fp32 compute: the exact kernels have no such variant
    code = C.nerf_mlp_dir_code(ray_dirs)
    sample_dirs = ray_dirs.index_select(0, ridx)
    rgb_a, den_a = C.nerf_mlp_forward(feats, sample_dirs, params, in_dim, 64, 4, True)
    rgb_b, den_b = C.nerf_mlp_forward(feats, None, params, in_dim, 64, 4, True, ray_code=(ridx, code))
    assert torch.equal(rgb_a, rgb_b) and torch.equal(den_a, den_b)
    gf_a, gp_a = C.nerf_mlp_backward(feats, sample_dirs, params, g_rgb, g_den, in_dim, 64, 4, True)
    gf_b, gp_b = C.nerf_mlp_backward(feats, None, params, g_rgb, g_den, in_dim, 64, 4, True, ray_code=(ridx, code))
    assert gf_a.shape == (S, in_dim) and torch.equal(gf_a, gf_b)
    assert torch.equal(gp_a, gp_b)
    with pytest.raises(RuntimeError):                      # the exact fp32 kernels: no per-ray variant, the library says so
        C.nerf_mlp_forward(feats.float(), None, params, in_dim, 64, 4, False, ray_code=(ridx, code))


def test_direct_step_covers_hidden_128_under_amp(monkeypatch):
    """hidden_dim 128 (the reference's best published nerf_hash row) through the direct-issue step - the wide decoder kernels compute
    in bf16, so the direct path takes it under amp only - against the modular step from the same state: same sample count, loss
    and gradients."""
    import copy
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    nef, _, _ = _build_pair(lods=16, hidden=128)
    nef2 = copy.deepcopy(nef)
    o, d = make_rays(500, 195)
    jit = cuda(np.random.default_rng(196).uniform(size=(500, 96)).astype(np.float32))
    gts = cuda(np.random.default_rng(197).uniform(size=(500, 3)).astype(np.float32))
    rays = Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0)
    assert MultiviewTrainStep(Pipeline(copy.deepcopy(nef), PackedRFTracer(raymarch_type='ray', num_steps=96)), prune_every=-1,
                              enable_amp=False)._direct is None                    # fp32: stays modular
    tr1 = MultiviewTrainStep(Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=96, bg_color=(0, 0, 0))), prune_every=-1, enable_amp=True)
    assert tr1._direct is not None
    monkeypatch.setenv("WISP_DIRECT_STEP", "0")
    tr2 = MultiviewTrainStep(Pipeline(nef2, PackedRFTracer(raymarch_type='ray', num_steps=96, bg_color=(0, 0, 0))), prune_every=-1, enable_amp=True)
    assert tr2._direct is None
    grads = {}
    for name, tr in (("direct", tr1), ("modular", tr2)):
        def snap(tr=tr, name=name):
            grads[name] = tr.flat.grad.clone()
            tr.flat.grad.zero_()
        tr.optimizer_step = snap
    l1, s1 = tr1.step(rays, gts, jitter=jit)
    l2, s2 = tr2.step(rays, gts, jitter=jit)
    assert s1 == s2 and abs(float(l1) - float(l2)) <= 1e-6 * max(1.0, abs(float(l2)))
    g1, g2 = grads["direct"].cpu().numpy(), grads["modular"].cpu().numpy()
    np.testing.assert_allclose(g1, g2, rtol=0, atol=1e-5 * float(np.abs(g2).max()))


@pytest.mark.parametrize("amp", [False, True])
def test_direct_step_equals_modular_step(amp, monkeypatch):
    """The direct-issue step of MultiviewTrainStep (same launches, no module / autograd plumbing) against the modular
    Pipeline.forward + autograd step: same loss and same gradients, and the same training trajectory."""
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    import copy
    nef, _, _ = _build_pair(lods=16)
    nef2 = copy.deepcopy(nef)
    o, d = make_rays(500, 191)
    jit = cuda(np.random.default_rng(192).uniform(size=(500, 96)).astype(np.float32))
    gts = cuda(np.random.default_rng(193).uniform(size=(500, 3)).astype(np.float32))
    rays = Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0)
    tr1 = MultiviewTrainStep(Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=96, bg_color=(0, 0, 0))),
                             prune_every=-1, enable_amp=amp)
    assert tr1._direct is not None
    monkeypatch.setenv("WISP_DIRECT_STEP", "0")
    tr2 = MultiviewTrainStep(Pipeline(nef2, PackedRFTracer(raymarch_type='ray', num_steps=96, bg_color=(0, 0, 0))),
                             prune_every=-1, enable_amp=amp)
    assert tr2._direct is None
    # 1. same gradients: run one step with the optimizer replaced by a snapshot of the flat gradient buffer
    grads = {}
    for name, tr in (("direct", tr1), ("modular", tr2)):
        def snap(tr=tr, name=name):
            grads[name] = tr.flat.grad.clone()
            tr.flat.grad.zero_()
        tr.optimizer_step = snap
    l1, s1 = tr1.step(rays, gts, jitter=jit)
    l2, s2 = tr2.step(rays, gts, jitter=jit)
    assert s1 == s2 and abs(float(l1) - float(l2)) <= 1e-6 * max(1.0, abs(float(l2)))      # different summation order
    g1, g2 = grads["direct"].cpu().numpy(), grads["modular"].cpu().numpy()
    scale = float(np.abs(g2).max())
    # (this small batch takes the float-atomic scatter path, whose add order varies from run to run)
    np.testing.assert_allclose(g1, g2, rtol=0, atol=(2e-6 if not amp else 1e-5) * scale)
    # 2. same trajectory: real optimisation steps, losses stay together
    del tr1.optimizer_step, tr2.optimizer_step
    for _ in range(3):
        l1, s1 = tr1.step(rays, gts, jitter=jit)
        l2, s2 = tr2.step(rays, gts, jitter=jit)
        assert s1 == s2 and tr1.num_rays == tr2.num_rays
        # (Adam with eps = 1e-16 turns add-order noise on near-zero gradient entries into O(lr) parameter differences)
        assert abs(float(l1) - float(l2)) <= 1e-3 * max(1.0, abs(float(l2)))


@pytest.mark.parametrize("amp", [False, True])
@pytest.mark.parametrize("march,steps", [("voxel", 6), ("uniform", 96), ("ray", 96)])
@pytest.mark.parametrize("kind", ["hash", "octree", "codebook"])
def test_direct_step_equals_modular_step_every_march_and_grid(kind, march, steps, amp, monkeypatch):
    """The direct-issue step for everything else the plugin surface offers: the 'voxel' and 'uniform' marches
    (octree_as.py:188-245, 311-374; BASELINE configs 4 and 5 march 'voxel') and the OctreeGrid / CodebookOctreeGrid fields
    (nerf_octree.yaml, nerf_codebook.yaml: 5 'sum' features, the codebook one without decoder biases, RMSprop) - against the
    modular Pipeline.forward + autograd step from the same state with the same in-kernel jitter seed: same sample count, loss,
    and every gradient in the flat buffer."""
    import copy
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.models.grids import CodebookOctreeGrid, OctreeGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    if kind == "hash":
        if march == "ray":
            pytest.skip("covered by test_direct_step_equals_modular_step")
        nef, _, _ = _build_pair(lods=16)
    else:
        blas, _ = _sparse_blas(5, 3000, 131)
        torch.manual_seed(3)
        if kind == "octree":
            grid = OctreeGrid(blas, feature_dim=5, num_lods=4, multiscale_type='sum', feature_std=0.5)
        else:
            grid = CodebookOctreeGrid(blas, feature_dim=5, num_lods=4, multiscale_type='sum', feature_std=0.7, codebook_bitwidth=4)
        nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1,
                                  bias=(kind == "octree")).to(DEV)
        with torch.no_grad():
            for n, p in nef.named_parameters():
                if 'decoder' in n:
                    p.mul_(2.0)
    nef2 = copy.deepcopy(nef)
    o, d = make_rays(400, 291)
    gts = cuda(np.random.default_rng(293).uniform(size=(400, 3)).astype(np.float32))
    rays = Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0)
    opt = dict(optimizer='rmsprop', eps=1e-8, weight_decay=0.0, grid_lr_weight=100.0, rgb_loss_type='l2') if kind == "codebook" else {}
    tr1 = MultiviewTrainStep(Pipeline(nef, PackedRFTracer(raymarch_type=march, num_steps=steps, bg_color=(1.0, 1.0, 1.0))),
                             prune_every=-1, enable_amp=amp, **opt)
    assert tr1._direct is not None and tr1._direct.hash_fast == (kind == "hash")
    monkeypatch.setenv("WISP_DIRECT_STEP", "0")
    tr2 = MultiviewTrainStep(Pipeline(nef2, PackedRFTracer(raymarch_type=march, num_steps=steps, bg_color=(1.0, 1.0, 1.0))),
                             prune_every=-1, enable_amp=amp, **opt)
    assert tr2._direct is None
    grads = {}
    for name, tr in (("direct", tr1), ("modular", tr2)):
        def snap(tr=tr, name=name):
            grads[name] = tr.flat.grad.clone()
            tr.flat.grad.zero_()
        tr.optimizer_step = snap
    torch.manual_seed(11)
    l1, s1 = tr1.step(rays, gts)
    torch.manual_seed(11)                                   # the marches draw their jitter seed from torch's generator
    l2, s2 = tr2.step(rays, gts)
    assert s1 == s2 > 1000 and abs(float(l1) - float(l2)) <= 2e-6 * max(1.0, abs(float(l2)))
    g1, g2 = grads["direct"].cpu().numpy(), grads["modular"].cpu().numpy()
    scale = float(np.abs(g2).max())
    assert scale > 0 and np.abs(g2).astype(bool).mean() > 0.001
    np.testing.assert_allclose(g1, g2, rtol=0, atol=(2e-6 if not amp else 1e-5) * scale)
    # and real optimisation steps stay together
    del tr1.optimizer_step, tr2.optimizer_step
    for k in range(3):
        torch.manual_seed(20 + k)
        l1, s1 = tr1.step(rays, gts)
        torch.manual_seed(20 + k)
        l2, s2 = tr2.step(rays, gts)
        assert s1 == s2 and tr1.num_rays == tr2.num_rays
        assert abs(float(l1) - float(l2)) <= 1e-3 * max(1.0, abs(float(l2)))


@pytest.mark.parametrize("march,steps", [("voxel", 6), ("uniform", 96)])
def test_raytrace_issued_one_batch_ahead_changes_nothing(march, steps):
    """'voxel' / 'uniform' marches with the one-batch look-ahead (step(..., prefetch=next rays)): the next batch's cell
    intersection counts are issued a step early (OctreeAS.raytrace_begin) so the size read-back never drains the GPU.  Same
    seeds, same batches -> the same sample counts, first-step loss and gradient as without look-ahead; a state issued for
    another Rays object, another level or an octree a prune has since replaced is dropped, not used."""
    import copy
    from wisp.accelstructs import OctreeAS
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.models.grids import OctreeGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    blas, _ = _sparse_blas(5, 3000, 131)
    torch.manual_seed(3)
    grid = OctreeGrid(blas, feature_dim=5, num_lods=4, multiscale_type='sum', feature_std=0.5)
    nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1).to(DEV)
    nef2 = copy.deepcopy(nef)
    rng = np.random.default_rng(301)
    batches = []
    for k in range(5):
        o, d = make_rays(300, 310 + k)
        batches.append((Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0), cuda(rng.uniform(size=(300, 3)).astype(np.float32))))
    def trainer(n):
        return MultiviewTrainStep(Pipeline(n, PackedRFTracer(raymarch_type=march, num_steps=steps, bg_color=(1.0, 1.0, 1.0))),
                                  prune_every=-1, enable_amp=False)
    tr1, tr2 = trainer(nef), trainer(nef2)
    assert tr1._direct is not None
    out, grads = {1: [], 2: []}, {1: [], 2: []}
    for key, tr in ((1, tr1), (2, tr2)):
        def snap(tr=tr, key=key, inner=tr.optimizer_step):         # the gradient as the optimizer is about to see it
            grads[key].append(tr.flat.grad.clone())
            inner()
        tr.optimizer_step = snap
    for k, (rays, gts) in enumerate(batches):
        nxt = batches[k + 1][0] if k + 1 < len(batches) else None
        torch.manual_seed(40 + k)
        out[1].append(tr1.step(rays, gts, prefetch=nxt))
        if nxt is not None:
            assert tr1._direct._pending is not None and tr1._direct._pending["rays"] is nxt
        torch.manual_seed(40 + k)
        out[2].append(tr2.step(rays, gts))
        assert tr2._direct._pending is None
    # What the look-ahead could change is WHICH samples a step sees: counts and the first step's loss / gradient (same
    # parameters on both sides) are compared exactly resp. to rounding.  Every kernel of this step is order-free (the octree
    # grid's backward sums in fixed point), so the two runs are in fact bit-identical; the later steps are still only held to
    # what Adam (eps = 1e-16: a last-bit difference of a near-zero gradient flips a full +-lr step) guarantees.
    for (l1, s1), (l2, s2) in zip(out[1], out[2]):
        assert s1 == s2 > 500
    assert abs(float(out[1][0][0]) - float(out[2][0][0])) <= 1e-6 * max(1.0, abs(float(out[2][0][0])))
    g1, g2 = grads[1][0], grads[2][0]
    assert float((g1 - g2).abs().max()) <= 1e-6 * float(g2.abs().max()) and float(g2.abs().max()) > 0
    for (l1, _), (l2, _) in zip(out[1], out[2]):
        assert abs(float(l1) - float(l2)) <= 1e-3 * max(1.0, abs(float(l2)))
    assert torch.isfinite(tr1.flat.data).all()
    # a state that does not fit is ignored: other rays / other level / replaced octree
    b = grid.blas
    r0, r1 = batches[0][0], batches[1][0]
    ref = b.raytrace(r0, 4, with_exit=True)
    for begun in (b.raytrace_begin(r1, 4), b.raytrace_begin(r0, 3), OctreeAS(b.octree.clone()).raytrace_begin(r0, 4)):
        got = b.raytrace(r0, 4, with_exit=True, begun=begun)
        assert torch.equal(got.ridx, ref.ridx) and torch.equal(got.pidx, ref.pidx) and torch.equal(got.depth, ref.depth)
    got = b.raytrace(r0, 4, with_exit=True, begun=b.raytrace_begin(r0, 4))
    assert torch.equal(got.ridx, ref.ridx) and torch.equal(got.pidx, ref.pidx) and torch.equal(got.depth, ref.depth)


@pytest.mark.parametrize("kind", ["octree", "codebook"])
def test_octree_radiance_fields_fused_decoder_equals_module_decoder(kind):
    """nerf_octree.yaml / nerf_codebook.yaml shapes (5 'sum' grid features, hidden 64, 'voxel' march, white background):
    the pipeline with the fused HIP decoder (narrow-input path) against the same pipeline evaluating the decoder's torch
    modules - rendered colours and every parameter gradient (grid features, dictionaries, decoder weights)."""
    import copy
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.models.grids import CodebookOctreeGrid, OctreeGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    blas, _ = _sparse_blas(5, 3000, 131)
    torch.manual_seed(3)
    if kind == "octree":
        grid = OctreeGrid(blas, feature_dim=5, num_lods=4, multiscale_type='sum', feature_std=0.5)
    else:
        grid = CodebookOctreeGrid(blas, feature_dim=5, num_lods=4, multiscale_type='sum', feature_std=0.7, codebook_bitwidth=4)
    nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1,
                              bias=(kind == "octree")).to(DEV)
    nef.decoder_compute = 'fp32'
    with torch.no_grad():
        for n, p in nef.named_parameters():
            if 'decoder' in n:
                p.mul_(2.0)
    ref = copy.deepcopy(nef)
    ref.fused_decoder = False
    probe = torch.zeros(4, 5, device=DEV)
    assert nef._can_fuse(probe) and not ref._can_fuse(probe)
    o, d = make_rays(400, 132)
    rays = Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0)
    gts = cuda(np.random.default_rng(133).uniform(size=(400, 3)).astype(np.float32))
    rgbs = []
    for m in (nef, ref):
        torch.manual_seed(7)                               # same in-kernel jitter seed for both pipelines
        pipe = Pipeline(m, PackedRFTracer(raymarch_type='voxel', num_steps=4, bg_color=(1.0, 1.0, 1.0)))
        rb = pipe(rays=rays, channels=["rgb"])
        torch.nn.functional.smooth_l1_loss(rb.rgb, gts).backward()
        rgbs.append(rb.rgb.detach())
        assert pipe.tracer.get_prev_num_samples() > 1000
    np.testing.assert_allclose(rgbs[0].cpu().numpy(), rgbs[1].cpu().numpy(), atol=1e-4)
    checked = 0
    for (n1, p1), (n2, p2) in zip(sorted(nef.named_parameters()), sorted(ref.named_parameters())):
        assert n1 == n2 and (p1.grad is None) == (p2.grad is None)
        if p1.grad is None:
            continue
        scale = max(float(p2.grad.abs().max()), 1e-6)
        assert float((p1.grad - p2.grad).abs().max()) <= 3e-4 * scale + 1e-7, n1
        checked += 1
    assert checked >= 6


def test_sample_rays_one_launch_gather_equals_indexing():
    """SampleRays on GPU tensors gathers origins / dirs / rgb with ONE wisp_gather_rows launch: same rays as tensor indexing
    with the same random indices (ray_sampler.py:25-35), negative indices included at the C-ABI level."""
    from wisp.core import Rays
    from wisp.datasets import MultiviewBatch, SampleRays
    g = torch.Generator(device=DEV).manual_seed(9)
    n = 70001
    o = torch.randn(n, 3, device=DEV, generator=g); d = torch.randn(n, 3, device=DEV, generator=g)
    rgb = torch.rand(n, 3, device=DEV, generator=g); wide = torch.rand(n, 5, device=DEV, generator=g)
    idx = torch.randint(-n, n, (4099,), device=DEV, generator=g)
    got = _C().gather_rows(idx, [o, d, rgb, wide])
    for t, q in zip((o, d, rgb, wide), got):
        assert torch.equal(q, t[idx])
    batch = MultiviewBatch(rays=Rays(o, d, dist_min=1.0, dist_max=5.0), rgb=rgb)
    g1 = torch.Generator(device=DEV).manual_seed(3); g2 = torch.Generator(device=DEV).manual_seed(3)
    out = SampleRays(777)(batch, generator=g1)
    ridx = torch.randint(0, n, [777], device=DEV, generator=g2)
    assert torch.equal(out['rays'].origins, o[ridx]) and torch.equal(out['rays'].dirs, d[ridx]) and torch.equal(out['rgb'], rgb[ridx])
    assert out['rays'].dist_min == 1.0 and out['rays'].dist_max == 5.0


def test_step_with_prefetched_next_batch_is_the_same_step():
    """The one-batch look-ahead of MultiviewTrainStep.step (next batch's occupancy test issued early) changes the issue order
    only: same sample counts and the same loss trajectory as without it - also across a prune, which invalidates the
    look-ahead (the octree it was computed against is gone)."""
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    import copy
    nef0, _, _ = _build_pair(lods=16)
    batches = []
    for k in range(6):
        o, d = make_rays(600, 300 + k)
        gts = cuda(np.random.default_rng(400 + k).uniform(size=(600, 3)).astype(np.float32))
        batches.append((Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0), gts))
    logs = []
    for look_ahead in (False, True):
        nef = copy.deepcopy(nef0)
        tr = MultiviewTrainStep(Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=96, bg_color=(0, 0, 0))),
                                prune_every=3, enable_amp=False)
        assert tr._direct is not None
        torch.manual_seed(11)                              # the in-kernel jitter seeds come from torch's CPU generator
        log = []
        for k, (rays, gts) in enumerate(batches):
            nxt = batches[k + 1][0] if (look_ahead and k + 1 < len(batches)) else None
            loss, ns = tr.step(rays, gts, prefetch=nxt)
            log.append((ns, float(loss)))
        logs.append(log)
    for (n0, l0), (n1, l1) in zip(*logs):
        assert n0 == n1 and abs(l0 - l1) <= 1e-4 * max(1.0, abs(l0))


@pytest.mark.parametrize("amp", [False, True])
def test_dropin_sdf_trainer_class_on_the_gpu(amp):
    """app/nglod's regime: wisp.trainers.SDFTrainer (the mirror of sdf_trainer.py:32-124, pinned to the reference's method bodies on
    the host) over the real OctreeGrid + NeuralSDF, batches of 512 from an SDFTensorDataset in HBM, torch.optim.Adam built by
    init_optimizer - and, with enable_amp (nglod_octree.yaml:68), fp16 autocast around step() as BaseTrainer.iterate applies it.
    Against SDFTrainStep (the fused step) on the same batches: same losses, same parameter trajectory."""
    import copy
    from wisp.accelstructs import OctreeAS
    from wisp.datasets import SDFTensorDataset
    from wisp.models import Pipeline
    from wisp.models.grids import OctreeGrid
    from wisp.models.nefs import NeuralSDF
    from wisp.trainers import SDFTrainer, SDFTrainStep, ConfigSDFTrainer, ConfigAdam, ConfigDataloader
    rng = np.random.default_rng(141)
    P = rng.integers(0, 32, size=(4000, 3))
    blas = OctreeAS.from_quantized_points(cuda(P.astype(np.int16)), 5)
    torch.manual_seed(5)
    grid = OctreeGrid(blas, feature_dim=16, num_lods=3, multiscale_type='sum', feature_std=0.05)
    nef = NeuralSDF(grid, pos_embedder='none', position_input=True, hidden_dim=128, num_layers=1).to(DEV)
    twin = copy.deepcopy(nef)
    n = 4 * 512
    coords = cuda(((P[rng.integers(0, P.shape[0], n)] + rng.uniform(0.05, 0.95, (n, 3))) / 16 - 1).astype(np.float32))
    gts = cuda(rng.normal(size=(n, 1)).astype(np.float32) * 0.1)
    cfg = ConfigSDFTrainer(optimizer=ConfigAdam(lr=1e-3, eps=1e-15), dataloader=ConfigDataloader(batch_size=512), grid_lr_weight=2.0,
                           max_epochs=2, enable_amp=amp, only_last=True)
    tr = SDFTrainer(cfg, Pipeline(nef, None), SDFTensorDataset(coords, gts), device=DEV)
    assert tr.iterations_per_epoch == 4
    seen = []

    class _Recording:                                   # the loader's own batches, recorded for the fused step
        def __init__(self, inner):
            self.inner = inner
        def __len__(self):
            return len(self.inner)
        def __iter__(self):
            for b in self.inner:
                seen.append((b["coords"].clone(), b["sdf"].clone()))
                yield b
    tr.train_data_loader = _Recording(tr.train_data_loader)
    tr.is_optimization_running = True
    losses = []
    inner_step = tr.step
    def recording_step(data):                           # (the metrics are cleared at every epoch start, before the step)
        before = tr.tracker.metrics.total_loss
        inner_step(data)
        losses.append(tr.tracker.metrics.total_loss - before)
    tr.step = recording_step
    for _ in range(6):
        tr.iterate()
    assert len(losses) == 6 and tr.epoch == 2
    fused = SDFTrainStep(twin, lr=1e-3, eps=1e-15, grid_lr_weight=2.0, optimizer='adam')
    want = [float(fused.step(x, y)) * 512 for x, y in seen[:6]]
    assert all(x.shape == (512, 3) and x.is_cuda for x, _ in seen)
    np.testing.assert_allclose(losses, want, rtol=(2e-5 if not amp else 2e-2))
    if not amp:
        for (n1, p1), (n2, p2) in zip(sorted(nef.named_parameters()), sorted(twin.named_parameters())):
            _assert_same_adam_trajectory(p1, p2, n1, steps=6, max_lr=2e-3)
    else:
        assert all(torch.isfinite(p).all() for p in nef.parameters()) and losses[-1] < losses[0]


def test_sdf_train_step_from_captured_graph_equals_eager_steps():
    """SDFTrainStep.capture(): forward + loss + backward of the 512-coordinate step recorded once as a HIP graph and
    replayed (optimizer launch outside the graph) must walk the same trajectory as the eager step - same kernels, same
    arguments; only the float atomics of the feature gradient may add in another order."""
    import copy
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import OctreeGrid
    from wisp.models.nefs import NeuralSDF
    from wisp.trainers import SDFTrainStep
    rng = np.random.default_rng(141)
    P = rng.integers(0, 32, size=(4000, 3))
    blas = OctreeAS.from_quantized_points(cuda(P.astype(np.int16)), 5)
    torch.manual_seed(6)
    grid = OctreeGrid(blas, feature_dim=16, num_lods=3, multiscale_type='sum', feature_std=0.05)
    nef_a = NeuralSDF(grid, pos_embedder='none', position_input=True, hidden_dim=128, num_layers=1).to(DEV)
    nef_b = copy.deepcopy(nef_a)
    eager = SDFTrainStep(nef_a, lr=1e-3, eps=1e-15, grid_lr_weight=2.0)
    graph = SDFTrainStep(nef_b, lr=1e-3, eps=1e-15, grid_lr_weight=2.0).capture(512)
    before = {n: p.detach().clone() for n, p in nef_b.named_parameters()}
    for n, p in nef_a.named_parameters():                      # capturing (warm-up passes included) moved no parameter
        assert torch.equal(p.detach(), before[n]), n
    for it in range(6):
        cells = cuda(((P[rng.integers(0, P.shape[0], 512)] + rng.uniform(0.05, 0.95, (512, 3))) / 16 - 1).astype(np.float32))
        gts = cuda(rng.normal(size=(512, 1)).astype(np.float32) * 0.1)
        la, lb = eager.step(cells, gts), graph.step(cells, gts)
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(la))), it
    moved = 0.0
    for (n1, p1), (n2, p2) in zip(sorted(nef_a.named_parameters()), sorted(nef_b.named_parameters())):
        _assert_same_adam_trajectory(p1, p2, n1, steps=6, max_lr=2e-3)
        moved = max(moved, float((p2.detach() - before[n2]).abs().max()))
    assert moved > 1e-3                                        # the replayed steps did train
    # another batch size falls back to eager issue
    cells = cuda(((P[rng.integers(0, P.shape[0], 100)] + 0.5) / 16 - 1).astype(np.float32))
    assert torch.isfinite(graph.step(cells, cuda(np.zeros((100, 1), np.float32))))


def test_sdf_tracer_fused_iteration_equals_modular_marching(monkeypatch):
    """wisp_sdf_trace_step_fused (step + octree walk + multi-level trilinear + decoder in one launch per iteration) against
    the modular loop (sphere_trace_step kernel, then nef(...) through spc_query / trilinear / torch Linear modules) on a
    NeuralSDF that was actually fitted to a sphere: same packs hit, same depths and positions up to the summation order of
    the decoder's dot products (rays whose decision sits on the convergence threshold are compared by count)."""
    from wisp.accelstructs import OctreeAS
    from wisp.core import Rays
    from wisp.models.grids import OctreeGrid
    from wisp.models.nefs import NeuralSDF
    from wisp.tracers import PackedSDFTracer
    from wisp.trainers import SDFTrainStep
    level = 5
    idx = np.stack(np.meshgrid(*[np.arange(32)] * 3, indexing='ij'), -1).reshape(-1, 3)
    ctr = (idx + 0.5) / 16 - 1
    P = idx[np.abs(np.linalg.norm(ctr, axis=1) - 0.55) < 0.15]
    blas = OctreeAS.from_quantized_points(torch.from_numpy(P).short().to(DEV), level)
    torch.manual_seed(11)
    grid = OctreeGrid(blas, feature_dim=16, num_lods=3, multiscale_type='sum', feature_std=0.01)
    nef = NeuralSDF(grid, pos_embedder='none', position_input=True, hidden_dim=128, num_layers=1).to(DEV)
    tr = SDFTrainStep(nef, lr=3e-3, grid_lr_weight=10.0)
    g = torch.Generator(device=DEV).manual_seed(12)
    cells = cuda(P.astype(np.float32))
    for _ in range(300):
        pick = torch.randint(0, cells.shape[0], (2048,), device=DEV, generator=g)
        xs = (cells[pick] + torch.rand(2048, 3, device=DEV, generator=g)) / 16 - 1
        tr.step(xs, xs.norm(dim=-1, keepdim=True) - 0.55)
    o, d = make_rays(3000, 151, radius=2.5, spread=0.7)
    rays = Rays(cuda(o), cuda(d), dist_min=0.0, dist_max=6.0)
    tracer = PackedSDFTracer(num_steps=40, step_size=0.8, min_dis=0.0003)
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("WISP_SDF_FUSED", fused)
        assert (PackedSDFTracer._fused_field(nef, 2) is not None) == (fused == "1")
        outs.append(tracer(nef, rays=rays, channels=["depth", "hit"], lod_idx=2))
    a, b = outs
    hits_a, hits_b = a.hit.reshape(-1), b.hit.reshape(-1)
    assert int(hits_b.sum()) > 500                                   # the fitted field is a surface the rays find
    differ = int((hits_a != hits_b).sum())
    assert differ <= max(2, int(0.002 * hits_b.numel())), differ
    both = hits_a & hits_b
    # a ray may converge one iteration earlier on one side when its distance sits at the threshold: both stop within
    # min_dis of the surface, so depths agree to ~2 min_dis; the bulk is identical to rounding
    dd = (a.depth.reshape(-1)[both] - b.depth.reshape(-1)[both]).abs()
    assert float(dd.max()) <= 6e-4 and float(dd.median()) <= 1e-6
    assert float((a.xyz[both] - b.xyz[both]).abs().max()) <= 6e-4
    # and the surface found is the sphere the field was fitted to
    assert float((b.xyz[both].norm(dim=-1) - 0.55).abs().mean()) < 0.02


def test_hidden_128_pipeline_trains_through_the_fused_wide_decoder(monkeypatch):
    """nerf_hash with hidden_dim=128 (the reference's best row) under bf16 autocast: the trainer's modular path must reach the
    fused wide decoder (no nn.Linear launches) and the loss must go down.  (The direct-issue step takes this shape too:
    test_direct_step_covers_hidden_128_under_amp.)"""
    monkeypatch.setenv("WISP_DIRECT_STEP", "0")
    import synlego
    import wisp._C as C
    from wisp.accelstructs import OctreeAS
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    torch.manual_seed(0)
    cells = synlego.occupied_cells(5, device=DEV)
    blas = OctreeAS.from_quantized_points(cells, 5)
    grid = HashGrid.from_geometric(blas, feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=1e-4, codebook_bitwidth=14,
                                   min_grid_res=8, max_grid_res=128)
    nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=128, num_layers=1, bias=True).to(DEV)
    pipe = Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=128, bg_color=(0.0, 0.0, 0.0)))
    tr = MultiviewTrainStep(pipe, prune_every=-1, enable_amp=True, lr=2e-3, grid_lr_weight=100.0)
    assert tr._direct is None
    o, d, gt = synlego.ray_bank(8192, seed=3, device=DEV)
    rays = Rays(o, d, dist_min=1.0, dist_max=5.0)
    C.TIMING_ALL = {}
    losses = [float(tr.step(rays, gt)[0]) for _ in range(40)]
    sink, C.TIMING_ALL = C.TIMING_ALL, None
    assert "wisp_nerf_mlp_fwd" in sink and "wisp_nerf_mlp_bwd" in sink and len(sink["wisp_nerf_mlp_bwd"]) == 40
    assert np.isfinite(losses).all() and losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])


@pytest.mark.parametrize("kind", ["codebook", "octree5", "octree16"])
def test_octree_fields_backward_is_bitwise_repeatable(kind):
    """the trilinear / codebook backward sums in 64-bit fixed point, so 50 launches on the same inputs - ray-
    ordered samples with long runs AND heavy corner sharing between rays - give the same bits 50 times, the per-level entry
    points agree with the all-levels launch to the fixed-point resolution, and nothing depends on what ran before."""
    from wisp.models.grids import CodebookOctreeGrid, OctreeGrid
    blas, oblas = _sparse_blas(5, 3000, 191)
    torch.manual_seed(6)
    F = 16 if kind == "octree16" else 5
    if kind == "codebook":
        grid = CodebookOctreeGrid(blas, feature_dim=F, num_lods=4, multiscale_type='sum', feature_std=0.7, codebook_bitwidth=4).to(DEV)
    else:
        grid = OctreeGrid(blas, feature_dim=F, num_lods=4, multiscale_type='sum', feature_std=0.5).to(DEV)
    rng = np.random.default_rng(192)
    leaf = oblas.level_points().astype(np.float32)
    n_cells = 6000
    cells = np.repeat(leaf[rng.integers(0, min(200, leaf.shape[0]), n_cells)], 16, axis=0)     # 96 000 samples over <= 200 cells
    coords = cuda(((cells + rng.uniform(0, 1, cells.shape)) / 32.0 * 2 - 1).astype(np.float32))
    L = 4
    levels = grid.active_lods[:L]
    chain = blas.query_chain(coords, levels[-1], grid.base_lod)
    trk = grid.trinkets.int().to(DEV)
    g = torch.randn(coords.shape[0], F, device=DEV) * torch.logspace(-3, 2, coords.shape[0], device=DEV)[:, None]
    C = _C()
    if kind == "codebook":
        def run():
            gl, gd = C.codebook_trilinear_multi_backward(coords, chain, blas.points, trk, [f.detach() for f in grid.features[:L]],
                                                         [d.detach() for d in grid.dictionary[:L]], g, levels, True)
            return gl + gd
    else:
        def run():
            return C.spc_trilinear_multi_backward(coords, chain, blas.points, trk, g, [tuple(f.shape) for f in grid.features[:L]],
                                                  levels, True)
    first = run()
    assert all(float(t.abs().max()) > 0 for t in first)
    for rep in range(50):
        if rep % 10 == 3:                                   # something else in between, on the same scratch
            C.spc_trilinear_multi_backward(coords[:999], chain[:999], blas.points, trk, g[:999, :F] * 7,
                                           [tuple(f.shape) for f in grid.features[:L]], levels, True) if kind != "codebook" else None
        again = run()
        assert all(torch.equal(a, b) for a, b in zip(again, first)), rep
    # the per-level entry points (another scale: each level's own largest product) agree to the fixed-point resolution
    for i in range(L):
        cell_i = chain[:, i].contiguous()
        if kind == "codebook":
            gl, gd = C.codebook_trilinear_backward(coords.view(-1, 1, 3), cell_i, blas.points, trk, grid.features[i].detach(),
                                                   grid.dictionary[i].detach(), g.view(-1, 1, F), levels[i])
            for a, b in ((gl, first[i]), (gd, first[L + i])):
                assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
        else:
            gi = C.spc_trilinear_backward(coords.view(-1, 1, 3), cell_i, blas.points, trk, g.view(-1, 1, F), tuple(grid.features[i].shape),
                                          levels[i])
            assert float((gi - first[i]).abs().max()) <= 2e-6 * float(first[i].abs().max())


@pytest.mark.parametrize("batch", [512, 37, 5000])
def test_fused_sdf_train_step_equals_the_modular_launches_and_repeats_bitwise(batch, monkeypatch):
    """wisp_sdf_train_step (walk + lookups + decoder forward / backward per sample, fixed-order weight-gradient sums, order-free
    corner scatter: four launches) against the modular step it replaces (query, multi-level lookup, small decoder, torch loss,
    their backward passes) from the same parameters: same loss, same gradient in every parameter - and the same bits when run
    again."""
    import copy
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import OctreeGrid
    from wisp.models.nefs import NeuralSDF
    from wisp.trainers import SDFTrainStep
    rng = np.random.default_rng(240 + batch)
    P = rng.integers(0, 32, size=(4000, 3))
    blas = OctreeAS.from_quantized_points(cuda(P.astype(np.int16)), 5)
    torch.manual_seed(7)
    grid = OctreeGrid(blas, feature_dim=16, num_lods=4, multiscale_type='sum', feature_std=0.05)
    nef_a = NeuralSDF(grid, pos_embedder='none', position_input=True, hidden_dim=128, num_layers=1).to(DEV)
    nef_b = copy.deepcopy(nef_a)
    fused = SDFTrainStep(nef_a, lr=1e-3, eps=1e-15, grid_lr_weight=2.0)
    assert fused._fused_field() is not None
    monkeypatch.setenv("WISP_SDF_TRAIN_FUSED", "0")
    modular = SDFTrainStep(nef_b, lr=1e-3, eps=1e-15, grid_lr_weight=2.0)
    assert modular._fused_field() is None
    inside = ((P[rng.integers(0, P.shape[0], batch)] + rng.uniform(0.02, 0.98, (batch, 3))) / 16 - 1).astype(np.float32)
    inside[::11] = rng.uniform(-1.2, 1.2, (inside[::11].shape[0], 3))                 # some outside every cell / the unit cube
    coords = cuda(inside)
    gts = cuda(rng.normal(size=(batch, 1)).astype(np.float32) * 0.1)
    la = fused._forward_backward(coords, gts)
    lb = modular._forward_backward(coords, gts)
    assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb)))
    ga, gb = fused.flat.grad.clone(), modular.flat.grad.clone()
    for (n1, p1), (n2, p2) in zip(nef_a.named_parameters(), nef_b.named_parameters()):
        sc = max(float(p2.grad.abs().max()), 1e-12)
        margin(f"fused sdf step grad {n1} B={batch}", float((p1.grad - p2.grad).abs().max()), 2e-5 * sc)
    assert float(gb.abs().max()) > 0
    for _ in range(5):                                        # same inputs, same bits
        fused.flat.grad.zero_()
        l2 = fused._forward_backward(coords, gts)
        assert torch.equal(fused.flat.grad, ga) and float(l2) == float(la)
    # and it trains: a few real steps bring the loss down
    losses = [float(fused.step(coords, gts)) for _ in range(30)]
    assert losses[-1] < 0.7 * losses[0]


@pytest.mark.parametrize("amp", [False, True])
def test_tracer_as_one_autograd_node_equals_the_modular_graph_bit_for_bit(amp, monkeypatch):
    """PackedRFTracer.trace with the shipped NeRF shape runs lookup, decoder and compositing as ONE autograd node
    (wisp/tracers/_fused_trace.py: the three Functions' own forward / backward bodies back to back) - against the modular graph
    (WISP_FUSED_TRACE=0) on an unchanged-trainer style model (separate parameter tensors, torch.optim): same rgb / alpha / depth /
    hit, same gradient in every parameter, bit for bit, without and with fp16 autocast + a loss scale."""
    import copy
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.tracers import PackedRFTracer, _fused_trace
    nef, _, _ = _build_pair(lods=16)
    nef2 = copy.deepcopy(nef)
    o, d = make_rays(500, 391)
    jit = cuda(np.random.default_rng(392).uniform(size=(500, 96)).astype(np.float32))
    gts = cuda(np.random.default_rng(393).uniform(size=(500, 3)).astype(np.float32))
    rays = Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0)
    outs = {}
    for name, model, enabled in (("fused", nef, True), ("modular", nef2, False)):
        monkeypatch.setattr(_fused_trace, "ENABLED", enabled)
        pipe = Pipeline(model, PackedRFTracer(raymarch_type='ray', num_steps=96, bg_color=(0.1, 0.2, 0.3)))
        with torch.autocast('cuda', enabled=amp):
            rb = pipe(rays=rays, channels=["rgb", "alpha", "depth", "hit"], jitter=jit)
            loss = torch.nn.functional.smooth_l1_loss(rb.rgb, gts, reduction='none').mean() + 0.1 * ((1.0 - rb.alpha) ** 2).mean() \
                + 1e-3 * rb.depth.mean()
        node_names = set()
        stack = [loss.grad_fn]
        while stack:
            fn = stack.pop()
            if fn is None or fn in node_names:
                continue
            node_names.add(fn)
            stack.extend(f for f, _ in fn.next_functions)
        kinds = {type(f).__name__ for f in node_names}
        (loss * (65536.0 if amp else 1.0)).backward()
        outs[name] = (rb, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, kinds)
    assert any("_FusedTrace" in k for k in outs["fused"][2]) and not any("_FusedTrace" in k for k in outs["modular"][2])
    assert not any(k.startswith(("HashGridInterpolate", "_FusedDecoder", "_Composite")) for k in outs["fused"][2])
    a, b = outs["fused"][0], outs["modular"][0]
    for ch in ("rgb", "alpha", "depth", "hit"):
        assert torch.equal(getattr(a, ch), getattr(b, ch)), ch
    ga, gb = outs["fused"][1], outs["modular"][1]
    assert set(ga) == set(gb) and len(ga) == 11
    # (this batch is below the binned backward's size: the table gradient goes through float atomics on both sides, whose order is
    #  free - everything else is bit-identical)
    for n in ga:
        if 'codebook' in n:
            sc = float(gb[n].abs().max())
            assert float((ga[n] - gb[n]).abs().max()) <= 2e-6 * sc, n
        else:
            assert torch.equal(ga[n], gb[n]), n


def test_hidden_128_pipelined_dw_kernel_equals_the_barrier_per_stage_kernel_bitwise(tmp_path):
    """wide_dw2_kernel (producer / consumer waves, barriers paired by count) against the one-role kernel it replaced
    (WISP_WIDE_DW=1; read once per process, hence the child): the same sums in the same order - parameter gradients and
    feature gradients bit for bit, at a sample count that leaves ragged rounds."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("", "1"):
        out = str(tmp_path / f"dw{mode or 0}.pt")
        env = dict(os.environ, WISP_WIDE_DW=mode)
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "bench_wide_dw.py"), out, "70001,300007"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        outs[mode] = torch.load(out)
    for S in outs[""]:
        gp_a, gf_a = outs[""][S]
        gp_b, gf_b = outs["1"][S]
        assert float(gp_a.abs().max()) > 0 and torch.equal(gp_a, gp_b) and torch.equal(gf_a, gf_b), S


@pytest.mark.parametrize("direct", [True, False])
def test_gradient_accumulation_equals_one_step_over_the_union_of_the_micro_batches(direct, monkeypatch):
    """MultiviewTrainStep.accumulate (scripts/time_to_psnr.py emulates the 8-GPU weak-scaling batch with it): two micro-batches of R
    rays each, accumulated, then applied with grad_accum_steps = 2 - the mean of the two per-batch mean losses - against ONE
    step over the 2 R rays (the mean over the union): same gradient up to summation order, same parameters after the
    optimizer step; and the optimizer of the hash table is NOT folded into the backward while a gradient is being accumulated."""
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    import copy
    if not direct:
        monkeypatch.setenv("WISP_DIRECT_STEP", "0")
    nef, _, _ = _build_pair(lods=16)
    nef2 = copy.deepcopy(nef)
    o, d = make_rays(600, 291)
    jit = cuda(np.random.default_rng(292).uniform(size=(600, 96)).astype(np.float32))
    gts = cuda(np.random.default_rng(293).uniform(size=(600, 3)).astype(np.float32))
    mk = lambda a, b: Rays(cuda(o[a:b]), cuda(d[a:b]), dist_min=1.0, dist_max=5.0)
    tr1 = MultiviewTrainStep(Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=96, bg_color=(0, 0, 0))), prune_every=-1)
    tr2 = MultiviewTrainStep(Pipeline(nef2, PackedRFTracer(raymarch_type='ray', num_steps=96, bg_color=(0, 0, 0))), prune_every=-1)
    assert (tr1._direct is not None) == direct
    tr1.grad_accum_steps = 2
    assert tr1._fused_update_args() is None
    grads = {}
    for name, tr in (("accum", tr1), ("union", tr2)):
        inner = tr.optimizer_step

        def snap(*a, tr=tr, name=name, inner=inner, **kw):
            grads[name] = tr.flat.grad.clone()
            return inner(*a, **kw)
        tr.optimizer_step = snap
    la, sa = tr1.accumulate(mk(0, 300), gts[:300], jitter=jit[:300])
    lb, sb = tr1.step(mk(300, 600), gts[300:], jitter=jit[300:])
    lu, su = tr2.step(mk(0, 600), gts, jitter=jit)
    assert sa + sb == su
    assert abs(0.5 * (float(la) + float(lb)) - float(lu)) <= 1e-6 * max(1.0, abs(float(lu)))
    g1, g2 = grads["accum"].cpu().numpy() * 0.5, grads["union"].cpu().numpy()      # (the optimizer divides by grad_accum_steps)
    np.testing.assert_allclose(g1, g2, rtol=0, atol=2e-6 * float(np.abs(g2).max()))
    tr1.wait_for_parameters(); tr2.wait_for_parameters()
    p1, p2 = tr1.flat.data.cpu().numpy(), tr2.flat.data.cpu().numpy()
    # (Adam's first step moves every parameter with a non-zero gradient by ~lr whatever its size: compare where the gradient is
    #  clearly non-zero, i.e. where add-order noise cannot flip a sign)
    sure = np.abs(g2) > 1e-4 * float(np.abs(g2).max())
    np.testing.assert_allclose(p1[sure], p2[sure], rtol=0, atol=2e-5)
    assert float(tr1.flat.grad.abs().max()) == 0.0                                 # consumed and zeroed


def _native_pair(num_steps=96, loss='huber', bg=(0, 0, 0)):
    from wisp.models import Pipeline
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    import copy
    nef, _, _ = _build_pair(lods=16)
    nef2 = copy.deepcopy(nef)
    mk = lambda n: MultiviewTrainStep(Pipeline(n, PackedRFTracer(raymarch_type='ray', num_steps=num_steps, bg_color=bg)),
                                      prune_every=-1, enable_amp=True, target_sample_size=2 ** 17, rgb_loss_type=loss)
    return mk(nef), mk(nef2)


def _run_steps(tr, batches, gts, steps, seed):
    """`steps` optimisation steps with the one-batch look-ahead of bench.py's loop; -> losses, sample counts"""
    from wisp.core import Rays
    torch.manual_seed(seed)                                   # the jitter seeds follow torch's CPU generator (OctreeAS._draw_seed)
    rays = [Rays(cuda(o), cuda(d), dist_min=1.0, dist_max=5.0) for o, d in batches]
    losses, counts = [], []
    for i in range(steps):
        nxt = rays[(i + 1) % len(rays)] if i + 1 < steps else None
        loss, S = tr.step(rays[i % len(rays)], gts[i % len(rays)], prefetch=nxt)
        losses.append(loss.clone())
        counts.append(S)
    tr.wait_for_parameters()
    torch.cuda.synchronize()
    return [float(l) for l in losses], counts


def _close(l1, l2, rel=1e-5):
    return len(l1) == len(l2) and all(abs(a - b) <= rel * max(1.0, abs(b)) for a, b in zip(l1, l2))


def test_native_step_equals_the_python_issued_step():
    """csrc/train_step.hip (wisp_nerf_step_run: every launch of the nerf_hash.yaml step issued by one call into the library) against
    _DirectNeRFStep.run + reduce_and_update issuing the same entry points from Python: same jitter seeds, same kernels, same
    arguments.  Two Python-issued runs of this shape are not bit-identical themselves (the coarse table levels are flushed with
    float atomics by several workgroups), so the comparison is: identical sample counts and ray counts every step, the first
    step's loss bit for bit, its first moment (0.1 x the gradient) to 1e-5 of its largest entry, and the losses of six steps with
    the look-ahead count and the table's AdamW folded into the backward to 1e-5."""
    tr1, tr2 = _native_pair()
    assert tr1._native is not None
    tr2._native = None
    rng = np.random.default_rng(311)
    batches = [make_rays(3000, 312 + k) for k in range(3)]
    gts = [cuda(rng.uniform(size=(3000, 3)).astype(np.float32)) for _ in range(3)]
    l1, c1 = _run_steps(tr1, batches, gts, 1, seed=77)
    l2, c2 = _run_steps(tr2, batches, gts, 1, seed=77)
    assert l1 == l2 and c1 == c2
    m1, m2 = tr1.flat.exp_avg, tr2.flat.exp_avg
    assert float(m2.abs().max()) > 0 and float((m1 - m2).abs().max()) <= 1e-5 * float(m2.abs().max())
    assert torch.equal(tr1.flat.shadow, tr1.flat.data[slice(*tr1.flat.ranges["grid"])].bfloat16())       # the bf16 copy follows the master
    l1, c1 = _run_steps(tr1, batches, gts, 6, seed=78)
    l2, c2 = _run_steps(tr2, batches, gts, 6, seed=78)
    assert tr1._native.steps == 7 and tr1._native.fallbacks == 0, (tr1._native.steps, tr1._native.fallbacks)
    assert c1 == c2 and min(c1) > 8192, (c1, c2)
    assert _close(l1, l2), (l1, l2)
    assert tr1.opt_steps == tr2.opt_steps == 7 and tr1.num_rays == tr2.num_rays
    assert float(tr1.flat.grad.abs().max()) == 0.0 and float(tr2.flat.grad.abs().max()) == 0.0


@pytest.mark.parametrize("loss,bg", [("l2", (1.0, 1.0, 1.0)), ("l1", (0.1, 0.2, 0.3))])
def test_native_step_other_losses_and_backgrounds(loss, bg):
    """the loss kind and the background colour travel in the step's configuration: L2 over a white background (the V8 / VQAD
    command lines), L1 over a coloured one - first step's loss bit for bit, four steps to 1e-5"""
    tr1, tr2 = _native_pair(loss=loss, bg=bg)
    tr2._native = None
    rng = np.random.default_rng(341)
    batches = [make_rays(2500, 342 + k) for k in range(2)]
    gts = [cuda(rng.uniform(size=(2500, 3)).astype(np.float32)) for _ in range(2)]
    l1, c1 = _run_steps(tr1, batches, gts, 4, seed=81)
    l2, c2 = _run_steps(tr2, batches, gts, 4, seed=81)
    assert tr1._native.steps == 4 and c1 == c2 and l1[0] == l2[0] and _close(l1, l2), (l1, l2)


def test_time_to_psnr_runs_both_regimes_and_the_model_learns():
    """bench_quality.time_to_psnr (bench.py's `quality`, scripts/time_to_psnr.py) on a small budget: a plain regime and one with
    gradient accumulation - PSNR on held-out rays rises with the rays consumed, the bookkeeping adds up"""
    import bench_quality as bq
    import synlego
    dev = torch.device(DEV)
    train = synlego.ray_bank(1 << 17, seed=1000, device=dev)
    held = synlego.ray_bank(1 << 12, seed=7, device=dev)
    for regime in (dict(target=2 ** 17, accum=1), dict(target=2 ** 16, accum=2)):
        r = bq.time_to_psnr(dev, regime, train, held, ray_budget=300000, checkpoints=(100000, 299999), num_steps=512, prune_every=50)
        assert r["rays"] >= 300000 and r["micro_batches_per_step"] == regime["accum"] and len(r["curve"]) >= 2
        first, last = r["curve"][0], r["curve"][-1]
        assert last[0] > first[0] and last[2] > first[2] > 0.0
        assert last[3] > first[3] and last[3] > 18.0, r["curve"]                    # it learns
        assert set(r["psnr_at_rays"]) == {"100000", "299999"}


def test_native_step_hands_a_batch_it_cannot_hold_to_the_python_issued_step(monkeypatch):
    """More packed samples than the native step's buffers were sized for (WISP_ERR_CAPACITY): the batch is counted again from the
    same seed and taken by the Python-issued step; a run that alternates between the two (every other batch fits) stays on the
    trajectory of a run that never used the native step."""
    from wisp.trainers._native_step import NativeHashStep
    rng = np.random.default_rng(321)
    batches = [make_rays(3000, 322), make_rays(700, 323)]
    gts = [cuda(rng.uniform(size=(3000, 3)).astype(np.float32)), cuda(rng.uniform(size=(700, 3)).astype(np.float32))]
    tr0, tr2 = _native_pair()
    tr2._native = None
    l2, c2 = _run_steps(tr2, batches, gts, 6, seed=78)
    cap = (min(c2) + max(c2)) // 2                              # the 3000-ray batches overflow, the 700-ray ones fit
    monkeypatch.setattr(NativeHashStep, "_capacity", lambda self: (int(self.t.max_rays), int(cap)))
    l1, c1 = _run_steps(tr0, batches, gts, 6, seed=78)
    assert tr0._native.fallbacks == 3 and tr0._native.steps == 3, (tr0._native.fallbacks, tr0._native.steps)
    assert c1 == c2 and _close(l1, l2), (l1, l2)


def test_native_step_survives_a_prune_and_a_change_of_batch_size():
    """The handle borrows the octree's buffers: a prune replaces them (the handle is rebuilt, the batch counted ahead is counted
    again from its seed) and a larger target re-sizes the workspace - the run stays on the Python-issued one's trajectory."""
    tr1, tr2 = _native_pair()
    tr2._native = None
    rng = np.random.default_rng(331)
    batches = [make_rays(3000, 332 + k) for k in range(2)]
    gts = [cuda(rng.uniform(size=(3000, 3)).astype(np.float32)) for _ in range(2)]
    for tr in (tr1, tr2):
        _run_steps(tr, batches, gts, 3, seed=79)
        tr.prune()
        tr.target_sample_size = 2 ** 18
    l1, c1 = _run_steps(tr1, batches, gts, 4, seed=80)
    l2, c2 = _run_steps(tr2, batches, gts, 4, seed=80)
    assert tr1._native.steps == 7 and c1 == c2 and _close(l1, l2), (c1, c2, l1, l2)


def test_bench_line_on_the_gpu_has_the_contract_fields_and_is_alone_on_stdout():
    """`python bench.py` in a short form, as a subprocess on the real device: exactly ONE line on stdout (RCCL's banner of the
    one-rank group of dp_path_regime must not reach it), and in it the contract's fields - metric / value / unit / n_gpus / steps
    / warmup / ms_per_step / scaling / dtype / config.workload, `roofline` (bound, achieved, peak, unit, frac, traffic) - plus
    this repository's regimes: large_batch_regime with its own roofline, quality, dropin_regime, dp_path_regime."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--pretrain", "30",
           "--bank-rays", "131072", "--eval-rays", "4096", "--large-target-samples", "524288", "--quality-budget", "200000",
           "--dp-steps", "4", "--dropin-steps", "5", "--no-configs", "--no-cpu-baseline", "--no-pmc"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["dtype"] == "bf16" and d["vs_baseline"] is None
    assert d["config"]["target_samples_per_step"] == 2 ** 18 and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["rays_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s") and 0.0 < roof["frac"] < 1.5
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and "traffic" in roof
    assert set(roof["all_kernels"]) >= {"hashgrid_fwd", "hashgrid_bwd", "nerf_mlp_fwd", "nerf_mlp_bwd"}
    lg = d["large_batch_regime"]
    assert lg["target_samples_per_step"] == 524288 and abs(lg["lr_scale"] - 2 ** 0.5) < 1e-9 and lg["roofline"]["frac"] > 0
    q = d["quality"]
    assert "error" not in q and q["headline"]["optimizer_steps"] > 0 and q["large_batch_regime"]["final_psnr_db"] > 10.0
    assert d["dropin_regime"]["ms_per_step"] > 0
    dp = d["dp_path_regime"]
    assert "error" not in dp and dp["allreduce"]["ms_per_step"] > 0 and dp["sharded"]["comm"]["path"].startswith("reduce-scatter")
    assert "PROJECTION" in dp["allreduce"]["projected_n8"]["note"]
