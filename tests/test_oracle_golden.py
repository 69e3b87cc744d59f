"""CPU: pins the oracle restatement against the committed golden vectors (tests/golden/make_golden.py).
hashgrid_ref.npz / uniform_ref.npz come from the reference's own kernel bodies compiled for the host."""
import os

import numpy as np
import pytest
import torch

from oracle import hashgrid, raymarch, render, spc, ref_lib


@pytest.fixture(scope="module")
def hg(golden_dir):
    return np.load(os.path.join(golden_dir, "hashgrid_ref.npz"))


def test_hash_index_kats(hg):
    for (x, y, z, r), want in zip(hg["kat_in"], hg["kat_out"]):
        c = torch.tensor([[0.0, 0.0, 0.0]])
        dense = hashgrid.level_is_dense(int(r), 2 ** 19, 3)
        if dense:
            got = x + y * r + z * r * r
        else:
            got = ((x * 1) & 0xFFFFFFFF ^ (y * 2654435761) & 0xFFFFFFFF ^ (z * 805459861) & 0xFFFFFFFF) % 2 ** 19
        assert int(got) == int(want), (x, y, z, r)
    # dense iff res^3 < T with strict '<' : 80 dense (512000 < 524288), 101 hashed
    assert hashgrid.level_is_dense(80, 2 ** 19) and not hashgrid.level_is_dense(101, 2 ** 19)
    for (x, y, r), want in zip(hg["kat2_in"], hg["kat2_out"]):
        dense = hashgrid.level_is_dense(int(r), 2 ** 19, 2)
        got = x + y * r if dense else (((x * 1) & 0xFFFFFFFF) ^ ((y * 2654435761) & 0xFFFFFFFF)) % 2 ** 19
        assert int(got) == int(want)


def test_clamp_bound_rounding(hg):
    # float32(res-1-1e-5) == res-1 exactly for res >= 258
    for r, p in zip(hg["probe_res"], hg["probe"]):
        assert np.float32(r - 1 - 1e-5) == p
        assert (p == r - 1) == (r >= 258)


@pytest.mark.parametrize("dim", [3, 2])
def test_hashgrid_forward_matches_reference_kernels(hg, dim):
    s = str(dim)
    res, bw = [int(r) for r in hg["res" + s]], int(hg["bw" + s])
    out = hashgrid.hashgrid_forward(torch.from_numpy(hg["coords" + s]), torch.from_numpy(hg["table" + s]),
                                    torch.from_numpy(hg["begin" + s]), res, bw).numpy()
    assert np.array_equal(out, hg["feats" + s])          # bit-exact against the reference kernel


@pytest.mark.parametrize("dim", [3, 2])
def test_hashgrid_backward_matches_reference_kernels(hg, dim):
    s = str(dim)
    res, bw = [int(r) for r in hg["res" + s]], int(hg["bw" + s])
    g = hashgrid.hashgrid_backward(torch.from_numpy(hg["coords" + s]), torch.from_numpy(hg["grad" + s]),
                                   hg["table" + s].shape, torch.from_numpy(hg["begin" + s]), res, bw, torch.float64).numpy()
    np.testing.assert_allclose(g, hg["gtable" + s], rtol=0, atol=2e-5)   # reference adds sequentially in fp32


def test_hashgrid_grad_coords_matches_reference_kernels(hg, golden_dir):
    """grad_coords of hashgrid_interpolate_backward_cuda(require_grad_coords=True): the oracle restates the reference kernel's
    arithmetic as it is (hashgrid_interpolate_cuda.cu:163-196) - bit for bit against the kernel body compiled for the host, on the
    committed vectors and (where oracle/_ref is built) on a fresh case with hashed levels at T = 2^14; and it is NOT the analytic
    derivative of the lookup (first-level columns for every level, corner 6 for 5, no res / 2), which the test pins as well so that
    nobody "fixes" one side only."""
    gc = np.load(os.path.join(golden_dir, "hashgrid_gradcoords_ref.npz"))
    for s in ("3", "2"):
        res, bw = [int(r) for r in hg["res" + s]], int(hg["bw" + s])
        out = hashgrid.hashgrid_grad_coords(torch.from_numpy(hg["coords" + s]), torch.from_numpy(hg["grad" + s]),
                                            torch.from_numpy(hg["table" + s]), hg["begin" + s], res, bw).numpy()
        assert out.shape == gc["gcoords" + s].shape == (hg["coords" + s].shape[0], 3)
        assert np.array_equal(out, gc["gcoords" + s])
    assert not gc["gcoords2"].any() and float(np.abs(gc["gcoords3"]).max()) > 1.0      # 2-D: the flag is ignored (zeros)
    # the analytic gradient of the oracle's own forward, by autograd over a torch restatement of the blend: a different function
    res, bw = [int(r) for r in hg["res3"]], int(hg["bw3"])
    c = torch.from_numpy(hg["coords3"][6:200]).clone().requires_grad_(True)
    table, begin = torch.from_numpy(hg["table3"]), hg["begin3"]
    feats = []
    for l, r in enumerate(res):
        x = torch.clamp((c.double() * 0.5 + 0.5) * float(r), 0.0, float(np.float32(r - 1 - 1e-5)))
        pos = torch.floor(x).detach()
        f = (x - pos).float()
        _, idx = hashgrid.corner_setup(c.detach(), r, 2 ** bw)
        acc = 0
        for k in range(8):
            w = (f[:, 0] if k & 4 else 1 - f[:, 0]) * (f[:, 1] if k & 2 else 1 - f[:, 1]) * (f[:, 2] if k & 1 else 1 - f[:, 2])
            acc = acc + table[int(begin[l]) + idx[:, k]] * w[:, None]
        feats.append(acc)
    (torch.cat(feats, 1) * torch.from_numpy(hg["grad3"][6:200])).sum().backward()
    assert float((c.grad - torch.from_numpy(gc["gcoords3"][6:200])).abs().max()) > 0.5
    if ref_lib.available():
        rng = np.random.default_rng(77)
        res, bw = [8, 20, 33, 64, 100], 14
        _, begin = hashgrid.table_layout(res, 2 ** bw, 3)
        table = rng.uniform(-1, 1, (int(begin[-1]), 4)).astype(np.float32)
        coords = rng.uniform(-1.05, 1.05, (700, 3)).astype(np.float32)
        go = rng.normal(size=(700, len(res) * 4)).astype(np.float32)
        want = ref_lib.hashgrid_grad_coords(coords, go, table, begin, res, bw)
        got = hashgrid.hashgrid_grad_coords(torch.from_numpy(coords), torch.from_numpy(go), torch.from_numpy(table), begin, res, bw).numpy()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("pb", [0, 1])
def test_hashgrid_query_matches_reference_kernels(golden_dir, pb):
    """oracle.hashgrid.hashgrid_query / _backward against the reference's own corner-query kernels (hashgrid_query_cuda.cu)."""
    q = np.load(os.path.join(golden_dir, "hashgrid_query_ref.npz"))
    res, bw = [int(r) for r in q["res"]], int(q["bw"])
    coords = torch.from_numpy(q["coords"])
    tables = [torch.from_numpy(t) for t in q["tables"]]
    out = hashgrid.hashgrid_query(coords, tables, res, bw, pb).numpy()
    assert np.array_equal(out, q[f"feats_p{pb}"])                     # a gather: bit-exact
    g = hashgrid.hashgrid_query_backward(coords, torch.from_numpy(q[f"grad_p{pb}"]), res, bw, 2, pb)
    np.testing.assert_allclose(np.stack([t.numpy() for t in g]), q[f"gtables_p{pb}"], rtol=0, atol=2e-5)   # sequential fp32 adds


def test_uniform_sampler_matches_reference_kernel(golden_dir):
    u = np.load(os.path.join(golden_dir, "uniform_ref.npz"))
    got = raymarch.uniform_sample(int(u["scale"]), u["ridx"], u["depth"], u["insum"])
    assert np.array_equal(got["ridx"], u["out_ridx"])
    assert np.array_equal(got["depth_samples"], u["out_depth"])
    assert np.array_equal(got["boundary"], u["out_boundary"])


@pytest.mark.parametrize("name", ["a", "b"])
def test_depth_bound_search_matches_reference_kernel(golden_dir, name):
    """oracle.sdf.find_depth_bound against the reference kernel body (render/find_depth_bound_cuda.cu:16-45, host build):
    inside / before / past-the-pack queries, finished packs, a finished right neighbour, and the last pack, whose bound is
    num_packs (so it finds nothing once its index is >= num_packs)."""
    from oracle import sdf
    g = np.load(os.path.join(golden_dir, "depth_bound_ref_%s.npz" % name))
    assert np.array_equal(sdf.find_depth_bound(g["query"], g["curr"], g["depth"]), g["out"])
    assert (g["out"] < 0).any() and (g["out"] >= 0).any() and (g["curr"] < 0).any()
    if name == "a":                                   # the last-pack quirk is exercised: valid start index, nothing found
        assert g["curr"][-1] >= len(g["curr"]) and g["out"][-1] == -1


def test_spc_builders_match_reference_function_bodies(golden_dir):
    """oracle.spc.pointcloud_to_octree / dilate_points against outputs of the reference's own function bodies
    (ops/spc/conversions.py:15-48, processing.py:13-47; tests/golden/make_golden.py): 0 / 1 / 2 dilation rounds, per-cell attribute
    means in morton order, and single cells - an interior cell grows into 23 neighbours (no centre, no -x-y / -x-z / -y-z edge)."""
    g = np.load(os.path.join(golden_dir, "spc_builders_ref.npz"))
    for level, rounds in g["cases"]:
        assert np.array_equal(spc.pointcloud_to_octree(g["cloud"], int(level), dilate=int(rounds)), g[f"octree_l{level}_d{rounds}"])
    tree, mean = spc.pointcloud_to_octree(g["cloud"], 5, attributes=g["attributes"])
    assert np.array_equal(tree, g["att_octree_l5"])
    np.testing.assert_allclose(mean, g["att_mean_l5"], atol=1e-6, rtol=0)
    for i, cell in enumerate(g["cells"]):
        assert np.array_equal(spc.dilate_points(cell[None], 5), g[f"dilated_{i}"])
    assert g["dilated_0"].shape[0] == 23 and g["cells"][0].tolist() not in g["dilated_0"].tolist()


def test_ray_generation_matches_reference_function_bodies(golden_dir):
    """oracle.raygen against outputs of the reference's generate_pinhole_rays / generate_ortho_rays function bodies
    (ops/raygen/raygen.py:40-119): 40 x 24 image, off-centre principal point."""
    from oracle import raygen
    g = np.load(os.path.join(golden_dir, "raygen_ref.npz"))
    W, H = int(g["width"]), int(g["height"])
    py, px = raygen.centered_pixel_coords(W, H)
    assert np.array_equal(py, g["pixel_y"]) and np.array_equal(px, g["pixel_x"])
    o, d = raygen.generate_rays(px, py, False, float(g["x0"]), float(g["y0"]), W, H, float(g["tan_h"]), float(g["tan_v"]),
                                g["rotation"], g["translation"])
    np.testing.assert_allclose(o, g["pinhole_origins"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(d, g["pinhole_dirs"], atol=2e-6, rtol=0)
    sx = np.float32(float(g["fov_distance"])) * np.float32(W / H)
    o, d = raygen.generate_rays(px, py, True, 0.0, 0.0, W, H, sx, float(g["fov_distance"]), g["rotation"], g["translation"])
    np.testing.assert_allclose(o, g["ortho_origins"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(d, g["ortho_dirs"], atol=2e-6, rtol=0)


@pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_live_reference_kernels_nerf_hash_shape():
    """nerf_hash.yaml shape (L=16, T=2^19, res 16..512) on fresh random inputs, forward bit-exact."""
    res = [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512]
    rng = np.random.default_rng(5)
    _, begin = hashgrid.table_layout(res, 2 ** 19)
    table = rng.uniform(-0.1, 0.1, (int(begin[-1]), 2)).astype(np.float32)
    coords = rng.uniform(-1, 1, (2000, 3)).astype(np.float32)
    want = ref_lib.hashgrid_forward(coords, table, begin, res, 19)
    got = hashgrid.hashgrid_forward(torch.from_numpy(coords), torch.from_numpy(table), torch.from_numpy(begin), res, 19)
    assert np.array_equal(got.numpy(), want)


def _oracle_cells(scalars, res):
    """x and floor(x) the oracle's corner_setup holds (oracle/hashgrid.py: the double expression, rounded once)."""
    c = torch.from_numpy(np.ascontiguousarray(scalars, dtype=np.float32))
    x = ((c.double() * 0.5 + 0.5) * float(res)).float()
    x = torch.clamp(x, min=0.0, max=float(np.float32(res - 1 - 1e-5)))
    return x.numpy(), torch.floor(x).to(torch.int32).numpy()


def test_oracle_cell_arithmetic_matches_reference_golden_cells(golden_dir):
    """Portable pin: floor(x) (all 16 NGP levels) and x (four levels) of the reference's 3-D kernel - its own code, tapped - on
    the structured adversarial coordinates (every cell face of every level +- 3 ulp, 2^-149 .. 2^1 of either sign, zeros,
    +-1, beyond +-1): the oracle's restatement gives the same integers and the same floats, bit for bit."""
    g = np.load(os.path.join(golden_dir, "hashgrid_cells_ref.npz"))
    s = g["scalars"]
    assert s.size > 16000
    xl = {int(l): i for i, l in enumerate(g["x_levels"])}
    for l, res in enumerate(g["res"]):
        x, pos = _oracle_cells(s, int(res))
        assert np.array_equal(pos.astype(np.int16), g["pos"][l]), res
        if l in xl:
            assert np.array_equal(x.view(np.uint32), g["x"][xl[l]].view(np.uint32)), res


@pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_one_fma_cell_formula_equals_reference_kernel_on_1e8_adversarial_coordinates():
    """csrc/hashgrid.hip scales a coordinate with ONE fp32 fma, fma(res/2, c, res/2), where the reference evaluates
    float(res * (c * 0.5 + 0.5)) in double (hashgrid_interpolate_cuda.cu:40-42, hash_utils.cuh:107-112).  The argument that the
    two agree needs |c| >= 2^-18; below that it is a probability.  Here: the reference's own kernel code (tapped right after its
    floor) against libm's correctly rounded fmaf on > 10^8 (coordinate, level) pairs concentrated where they could differ -
    1.5 M log-uniform magnitudes in [2^-149, 2^-18) incl. denormals, every cell face of every level +- 3 ulp, powers of two,
    +-1 and beyond, plus uniform draws - for all 16 NGP resolutions: the scaled position and its integer cell are IDENTICAL,
    bit for bit, on every pair (so are the eight corner rows, which are integer functions of the cell).  The device side of
    the claim - the GPU evaluates that fma - is tests/test_gpu_0_parity.py::test_hashgrid_cells_*."""
    import adversarial as adv
    s = np.concatenate([adv.structured_scalars(), adv.random_scalars(1_500_000, 500_000, 100_000, seed=1)])
    pts = adv.points(s, seed=2)
    pairs = 0
    for res in adv.NGP_RES:
        x, pos = ref_lib.cells_3d(pts, res)
        fx, fpos = ref_lib.fma_cells(pts, res)
        assert np.array_equal(pos, fpos), res
        assert np.array_equal(x.view(np.uint32), fx.view(np.uint32)), res
        pairs += pts.size
    assert pairs >= 10 ** 8
    # the oracle (what every GPU parity test compares with) on the structured part
    st = adv.structured_scalars()
    for res in adv.NGP_RES:
        x, pos = ref_lib.cells_3d(np.stack([st, st, st], 1), res)
        ox, opos = _oracle_cells(st, res)
        assert np.array_equal(opos, pos[:, 0]) and np.array_equal(ox.view(np.uint32), x[:, 0].view(np.uint32))


def test_float_order_alternatives_rarely_move_a_raymarch_candidate(capsys):
    """scripts/float_order_bound.py on a slice of the flagship shape (512 rays x 2048 candidates against the SynLego level-7
    occupancy): of the orderings a different Kaolin build could use inside the leaves the oracle had to restate - an FMA'd sample
    position, 0.5f*(x+1)*res or (x+1)*(res/2) or a double-precision scaling in the point query, |x| = 1 counted as outside - only
    the FMA moves any candidate to another cell at all (about 2 in a million), and none of them changes an occupancy decision,
    i.e. ridx / boundary / the sample count.  The full 33.5 M-candidate table is profiles/r03_float_order_bound.txt."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("float_order_bound", os.path.join(root, "scripts", "float_order_bound.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv, sys_argv = ["float_order_bound.py", "512"], __import__("sys").argv
    __import__("sys").argv = argv
    try:
        mod.main()
    finally:
        __import__("sys").argv = sys_argv
    rows = {}
    for line in capsys.readouterr().out.splitlines()[2:]:
        name, count, rate = line.rsplit(None, 2)
        rows[name.strip()] = (int(count), float(rate))
    assert rows["q1 0.5f*(x+1)*res: cell"][0] == 0 and rows["q2 (x+1)*(res/2): cell"][0] == 0     # power-of-two scalings commute with rounding
    assert rows["pos_fma: cell"][1] < 2e-5 and rows["q3 double, one rounding: cell"][1] < 2e-5
    for k in ("pos_fma: occupancy", "q1: occupancy", "q2: occupancy", "q3: occupancy", "x == +-1 outside: occupancy"):
        assert rows[k][1] < 5e-6, (k, rows[k])


def test_spc_kats(golden_dir):
    k = np.load(os.path.join(golden_dir, "spc_kats.npz"))
    oc = spc.points_to_octree(np.array([[0, 0, 0], [3, 3, 3], [2, 1, 0]]), 2)
    assert np.array_equal(oc, k["sp_octree"]) and oc.tolist() == [145, 1, 4, 128]
    pts, pyr, ex = spc.octree_to_spc(oc)
    assert np.array_equal(pts, k["sp_points"]) and np.array_equal(pyr, k["sp_pyramid"]) and np.array_equal(ex, k["sp_exsum"])
    assert np.array_equal(spc.query(oc, ex, k["sp_q"], 2, with_parents=True), k["sp_q_pidx"])
    r = spc.raytrace(oc, pts, pyr, ex, k["rt_o"], k["rt_d"], 2, with_exit=True)
    assert np.array_equal(r[0], k["rt_ridx"]) and np.array_equal(r[1], k["rt_pidx"]) and np.array_equal(r[2], k["rt_depth"])
    rd = spc.raytrace(k["dn_octree"], k["dn_points"], k["dn_pyramid"], k["dn_exsum"], k["dn_o"], k["dn_d"], 2, with_exit=True)
    assert np.array_equal(rd[1], k["dn_pidx"]) and np.allclose(rd[2], k["dn_depth"])


def test_query_is_membership_and_morton_rank():
    rng = np.random.default_rng(0)
    P = rng.integers(0, 32, size=(300, 3))
    oc = spc.points_to_octree(P, 5)
    pts, pyr, ex = spc.octree_to_spc(oc)
    leaf = pts[pyr[1, 5]:pyr[1, 5] + pyr[0, 5]]
    assert (np.diff(spc.points_to_morton(leaf)) > 0).all()
    x = rng.uniform(-1, 1, (4000, 3)).astype(np.float32)
    q = spc.quantize_points(x, 5)
    rank = {tuple(p): i for i, p in enumerate(leaf.tolist())}
    pid = spc.query(oc, ex, x, 5)
    for i in range(x.shape[0]):
        want = rank.get(tuple(q[i].tolist()), None)
        assert pid[i] == (-1 if want is None else pyr[1, 5] + want)


def test_raytrace_equals_bruteforce_slab_float64():
    rng = np.random.default_rng(2)
    P = rng.integers(0, 16, size=(60, 3))
    oc = spc.points_to_octree(P, 4)
    pts, pyr, ex = spc.octree_to_spc(oc)
    leaf = pts[pyr[1, 4]:pyr[1, 4] + pyr[0, 4]].astype(np.float64)
    o = rng.normal(size=(150, 3)); o = (3 * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    d = rng.uniform(-0.5, 0.5, (150, 3)) - o; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ridx, pidx, depth = spc.raytrace(oc, pts, pyr, ex, o, d, 4, with_exit=True)
    for r in range(150):
        lo = -1 + 2 * leaf / 16; hi = lo + 2 / 16
        t0 = (lo - o[r].astype(np.float64)) / d[r].astype(np.float64); t1 = (hi - o[r].astype(np.float64)) / d[r].astype(np.float64)
        tn = np.minimum(t0, t1).max(1); tf = np.maximum(t0, t1).min(1)
        solid = tf > np.maximum(tn, 0) + 1e-6          # exclude grazing hits
        graze = np.abs(tf - np.maximum(tn, 0)) <= 1e-6
        want = [pyr[1, 4] + k for k in np.argsort(np.maximum(tn, 0)) if solid[k]]
        mine = [p for p in pidx[ridx == r].tolist() if not graze[p - pyr[1, 4]]]
        assert mine == want


def test_raymarch_ray_properties():
    rng = np.random.default_rng(4)
    blas_oct = spc.points_to_octree(rng.integers(0, 16, size=(500, 3)), 4)
    pts, pyr, ex = spc.octree_to_spc(blas_oct)
    o = rng.normal(size=(50, 3)); o = (3 * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    d = rng.uniform(-0.5, 0.5, (50, 3)) - o; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    jit = rng.uniform(size=(50, 128)).astype(np.float32)
    rm = raymarch.raymarch_ray(blas_oct, ex, o, d, 1.0, 5.0, 128, 4, jit)
    assert (np.diff(rm["ridx"]) >= 0).all()
    assert np.array_equal(rm["boundary"], spc.mark_pack_boundaries(rm["ridx"]))
    assert (spc.query(blas_oct, ex, rm["samples"], 4) >= 0).all()
    assert (rm["deltas"] > 0).all() and (rm["depth_samples"] >= 1.0).all() and (rm["depth_samples"] <= 5.0 + 4.0 / 128).all()
    # linspace emulation equals torch.linspace to 1 ulp
    assert np.abs(raymarch.linspace01(128) - torch.linspace(0, 1, 128).numpy()).max() < 1e-7


def test_composite_matches_python_loop_float64():
    rng = np.random.default_rng(9)
    lens = [5, 1, 0, 17, 3]
    ridx = np.concatenate([np.full(n, r) for r, n in enumerate(lens)]).astype(np.int64)
    S = ridx.shape[0]
    color = torch.from_numpy(rng.uniform(size=(S, 3))); dens = torch.from_numpy(rng.uniform(0, 30, size=(S, 1)))
    delt = torch.from_numpy(rng.uniform(0.001, 0.1, size=(S, 1))); dep = torch.from_numpy(rng.uniform(1, 5, size=(S, 1)))
    b = torch.from_numpy(spc.mark_pack_boundaries(ridx))
    out = render.composite(color, dens, delt, dep, torch.from_numpy(ridx), b, 5, (0.2, 0.3, 0.4))
    for r, n in enumerate(lens):
        T, acc, A, D = 1.0, np.zeros(3), 0.0, 0.0
        for i in np.nonzero(ridx == r)[0]:
            tau = float(dens[i] * delt[i]); w = T * (1 - np.exp(-tau)); T *= np.exp(-tau)
            acc += w * color[i].numpy(); A += w; D += w * float(dep[i])
        want = np.array([0.2, 0.3, 0.4]) * (1 - A) + acc
        np.testing.assert_allclose(out["rgb"][r].numpy(), want, atol=1e-12)
        np.testing.assert_allclose(out["alpha"][r].item(), A, atol=1e-12)
        np.testing.assert_allclose(out["depth"][r].item(), D, atol=1e-12)
        assert bool(out["hit"][r]) == (A > 0)


def test_octree_grid_oracle_properties():
    """Trilinear coefficients are a partition of unity; sampling exactly at a voxel corner returns that corner's
    feature; samples outside the tree give zeros; trinkets index the per-level dual block."""
    from oracle import octree_grid as og
    rng = np.random.default_rng(3)
    P = rng.integers(0, 8, size=(60, 3))
    oc = spc.points_to_octree(P, 3)
    pts, pyr, ex = spc.octree_to_spc(oc)
    pd, pyd = spc.make_dual(pts, pyr)
    tr, par = spc.make_trinkets(pts, pyr, pd, pyd)
    level = 3
    feats = torch.randn(int(pyd[0, level]) + 1, 4)
    leaf = pts[pyr[1, level]:pyr[1, level] + pyr[0, level]]
    x = torch.from_numpy(rng.uniform(-1, 1, (500, 3)).astype(np.float32))
    pidx = torch.from_numpy(spc.query(oc, ex, x.numpy(), level))
    w = og.trilinear_coeffs(x, torch.from_numpy(pts)[pidx.clamp(min=0)].long(), level)
    assert torch.allclose(w.sum(-1)[pidx >= 0], torch.ones(int((pidx >= 0).sum())), atol=1e-5)
    out = og.interpolate_trilinear(x[:, None], pidx, pts, tr, feats, level)
    assert float(out[pidx < 0].abs().max()) == 0.0 and (pidx < 0).any()
    # exactly at the min corner of a voxel: coefficient 0 is 1 -> corner feature 0 of that voxel
    k = 7
    c = torch.from_numpy((leaf[k].astype(np.float32) / 8.0) * 2 - 1)[None, None]
    got = og.interpolate_trilinear(c, torch.tensor([pyr[1, level] + k]), pts, tr, feats, level)
    assert torch.allclose(got[0, 0], feats[tr[pyr[1, level] + k, 0]], atol=1e-6)
    assert tr.max() <= pyd[0, 1:].max() and par[0] == -1


def test_raygen_oracle_geometry():
    """The ray-generation restatement on a look-at camera: unit directions, the image-centre ray runs from the eye to the
    look-at point, and pixel (0.5, 0.5) of a 90-degree camera sits at the (-1, +1) corner of the image plane."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kaolin-wisp_amd"))
    from oracle import raygen as oray
    from wisp.ops.raygen import LookAtCamera
    cam = LookAtCamera(eye=(0.0, 0.0, 3.0), at=(0.0, 0.0, 0.0), up=(0, 1, 0), fov=np.pi / 2, width=64, height=64)
    m = cam.view_matrix()[0].numpy()
    py, px = oray.centered_pixel_coords(64, 64)
    o, d = oray.generate_rays(px, py, False, 0.0, 0.0, 64, 64, cam.tan_half_fov('horizontal'), cam.tan_half_fov('vertical'),
                              m[:3, :3], m[:3, 3])
    np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-6)
    np.testing.assert_allclose(o, np.tile(np.array([[0, 0, 3.0]], np.float32), (64 * 64, 1)), atol=1e-6)
    corner = d[0] / -d[0, 2]                                   # pixel (0.5, 0.5): top-left
    np.testing.assert_allclose(corner[:2], [-(1 - 1 / 64), (1 - 1 / 64)], atol=1e-6)
    oc, dc = oray.generate_rays(np.array([32.0], np.float32), np.array([32.0], np.float32), False, 0.0, 0.0, 64, 64, 1.0, 1.0,
                                m[:3, :3], m[:3, 3])
    np.testing.assert_allclose(dc[0], [0, 0, -1], atol=1e-6)
