"""Generates the committed golden vectors.  Run from the repo root in the build container (needs /root/reference
for the hash-grid / lattice-sampler vectors, which come from the REFERENCE's own kernel bodies compiled for the host
by oracle/build_ref.sh):

    python tests/golden/make_golden.py

Outputs (small, committed):
  hashgrid_ref.npz   - inputs + outputs of the reference hashgrid forward/backward kernels (3-D and 2-D, float32),
                       hash-index known answers, the clamp bound probes of SURVEY.md Appendix B
  hashgrid_gradcoords_ref.npz - grad_coords of the same kernels with require_grad_coords = true, on hashgrid_ref.npz's inputs
  hashgrid_query_ref.npz - inputs + outputs of the reference hashgrid_query forward/backward kernels (probe_bitwidth 0 and 1)
  uniform_ref.npz    - inputs + outputs of the reference uniform_sample kernel
  depth_bound_ref_{a,b}.npz - inputs + outputs of the reference find_depth_bound kernel (SDF tracer)
  spc_builders_ref.npz / raygen_ref.npz - outputs of the reference's pointcloud_to_octree / dilate_points and ray-generation
                       FUNCTION BODIES (compiled from the reference files; Kaolin leaves restated by the oracle)
  hashgrid_cells_ref.npz - the cell arithmetic of the reference's 3-D kernel (its own code, tapped after hashgrid_interpolate_cuda.cu:43)
                       on the structured adversarial coordinates of tests/adversarial.py: floor(x) for all 16 NGP levels, x itself
                       for four of them
  spc_kats.npz       - hand-checkable SPC cases (dense level-2 tree, 3-point sparse tree, query / raytrace answers)
                       produced by oracle/spc.py and verified inside this script against brute force in float64
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hashgrid, ref_lib, spc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def hashgrid_vectors():
    assert ref_lib.available(), "run oracle/build_ref.sh first"
    rng = np.random.default_rng(1234)
    out = {}
    # ---- 3-D: 4 dense + 4 hashed levels, T = 2^10, F = 2
    res3, bw3 = [4, 6, 8, 10, 13, 16, 23, 32], 10
    _, begin3 = hashgrid.table_layout(res3, 2 ** bw3, 3)
    table3 = rng.uniform(-0.5, 0.5, (int(begin3[-1]), 2)).astype(np.float32)
    coords3 = rng.uniform(-1, 1, (512, 3)).astype(np.float32)
    coords3[:6] = [[1, 1, 1], [-1, -1, -1], [0, 0, 0], [1, -1, 0.5], [0.999999, -0.999999, 0.25], [1.25, -3.0, 0.1]]
    grad3 = rng.normal(size=(512, len(res3) * 2)).astype(np.float32)
    out.update(res3=np.array(res3), bw3=bw3, begin3=begin3, table3=table3, coords3=coords3, grad3=grad3,
               feats3=ref_lib.hashgrid_forward(coords3, table3, begin3, res3, bw3),
               gtable3=ref_lib.hashgrid_backward(coords3, grad3, table3, begin3, res3, bw3))
    # ---- 2-D (image app): table sized with coord_dim = 3 (main_image.py:63), F = 4
    res2, bw2 = [4, 8, 16, 32, 64, 128], 9
    _, begin2 = hashgrid.table_layout(res2, 2 ** bw2, 3)
    table2 = rng.uniform(-0.5, 0.5, (int(begin2[-1]), 4)).astype(np.float32)
    coords2 = rng.uniform(-1, 1, (256, 2)).astype(np.float32)
    grad2 = rng.normal(size=(256, len(res2) * 4)).astype(np.float32)
    out.update(res2=np.array(res2), bw2=bw2, begin2=begin2, table2=table2, coords2=coords2, grad2=grad2,
               feats2=ref_lib.hashgrid_forward(coords2, table2, begin2, res2, bw2),
               gtable2=ref_lib.hashgrid_backward(coords2, grad2, table2, begin2, res2, bw2))
    # ---- index KATs at the nerf_hash.yaml sizes (T = 2^19): dense levels, first hashed level, finest level
    kat_in = np.array([[1, 2, 3, 16], [15, 15, 15, 16], [79, 79, 79, 80], [1, 2, 3, 80], [1, 2, 3, 101], [100, 100, 100, 101],
                       [1, 2, 3, 128], [511, 0, 255, 512], [512, 512, 512, 512], [0, 0, 0, 512], [321, 5, 77, 322]],
                      dtype=np.int64)
    kat_out = np.array([ref_lib.hash_index_3d(x, y, z, r, 2 ** 19) for x, y, z, r in kat_in], dtype=np.int64)
    kat2_in = np.array([[1, 2, 16], [100, 7, 128], [700, 700, 724], [723, 1, 724]], dtype=np.int64)
    kat2_out = np.array([ref_lib.hash_index_2d(x, y, r, 2 ** 19) for x, y, r in kat2_in], dtype=np.int64)
    # clamp bound: float32(res - 1 - 1e-5) probes (Appendix B)
    probe_res = np.array([80, 101, 257, 258, 322, 406, 512], dtype=np.int64)
    probe = np.array([ref_lib.clamp(1e9, 0, r - 1 - 1e-5) for r in probe_res], dtype=np.float32)
    out.update(kat_in=kat_in, kat_out=kat_out, kat2_in=kat2_in, kat2_out=kat2_out, probe_res=probe_res, probe=probe)
    np.savez_compressed(os.path.join(OUT, "hashgrid_ref.npz"), **out)


def gradcoords_vectors():
    """grad_coords of the reference's backward kernels (require_grad_coords = true) on hashgrid_ref.npz's own inputs."""
    assert ref_lib.available(), "run oracle/build_ref.sh first"
    g = np.load(os.path.join(OUT, "hashgrid_ref.npz"))
    out = {}
    for s in ("3", "2"):
        res, bw = [int(r) for r in g["res" + s]], int(g["bw" + s])
        out["gcoords" + s] = ref_lib.hashgrid_grad_coords(g["coords" + s], g["grad" + s], g["table" + s], g["begin" + s], res, bw)
    assert float(np.abs(out["gcoords3"]).max()) > 0.1 and not out["gcoords2"].any()
    np.savez_compressed(os.path.join(OUT, "hashgrid_gradcoords_ref.npz"), **out)


def query_vectors():
    """hashgrid_query_ref.npz: the reference's corner-query kernels (hashgrid_query_cuda.cu) on two dense + three hashed levels,
    without and with probe slots."""
    assert ref_lib.available(), "run oracle/build_ref.sh first"
    rng = np.random.default_rng(4321)
    res, bw, F = [4, 7, 12, 20, 33], 10, 2
    coords = rng.uniform(-1, 1, (200, 3)).astype(np.float32)
    coords[:4] = [[1, 1, 1], [-1, -1, -1], [0, 0, 0], [1.5, -2.0, 0.3]]
    tables = [rng.uniform(-0.5, 0.5, (2 ** bw, F)).astype(np.float32) for _ in res]
    out = dict(res=np.array(res), bw=bw, coords=coords, tables=np.stack(tables))
    for pb in (0, 1):
        P = 2 ** pb
        grad = rng.normal(size=(200, 8, len(res), P, F)).astype(np.float32)
        out[f"feats_p{pb}"] = ref_lib.hashgrid_query(coords, tables, res, bw, pb)
        out[f"grad_p{pb}"] = grad
        out[f"gtables_p{pb}"] = np.stack(ref_lib.hashgrid_query_backward(coords, grad, res, bw, F, pb))
    np.savez_compressed(os.path.join(OUT, "hashgrid_query_ref.npz"), **out)


def uniform_vectors():
    rng = np.random.default_rng(7)
    V = 300
    ridx = np.sort(rng.integers(0, 40, V)).astype(np.int32)
    entry = rng.uniform(0, 3, V).astype(np.float32)
    depth = np.stack([entry, entry + rng.uniform(0, 0.2, V).astype(np.float32)], 1)
    scale = 37
    ia = np.ceil(np.float32(scale) * depth[:, 0]).astype(np.int32)
    ib = np.ceil(np.float32(scale) * depth[:, 1]).astype(np.int32)
    cnt = ib - ia
    keep = cnt != 0
    insum = spc.inclusive_sum(cnt[keep])
    got = ref_lib.uniform_sample(scale, ridx[keep], depth[keep], insum)
    np.savez_compressed(os.path.join(OUT, "uniform_ref.npz"), scale=scale, ridx=ridx[keep], depth=depth[keep],
                        insum=insum, out_ridx=got["ridx"], out_depth=got["depth_samples"], out_boundary=got["boundary"])


def depth_bound_vectors():
    """depth_bound_ref.npz: the SDF tracer's depth-bound search (render/find_depth_bound_cuda.cu) on ragged packs: queries
    inside an interval, in the gap before one, past the pack's last exit (the search then runs into the NEXT pack - the
    kernel bounds it by the neighbour's start index only), finished packs (-1), a pack whose right neighbour is finished
    (bound 0xFFFFFFFF; queries kept inside so the reference body terminates), and the last pack (bounded by num_packs)."""
    rng = np.random.default_rng(11)
    for name, P in (("a", 48), ("b", 7)):
        counts = rng.integers(1, 6, P)
        start = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
        M = int(counts.sum())
        depth = np.empty((M, 2), np.float32)
        for p in range(P):
            e = np.sort(rng.uniform(0.2, 3.0, 2 * counts[p]).astype(np.float32))
            depth[start[p]:start[p] + counts[p], 0] = e[0::2]
            depth[start[p]:start[p] + counts[p], 1] = e[1::2]
        cur = start.copy()
        dead = rng.random(P) < 0.2
        dead[0] = False
        cur[dead] = -1
        kind = rng.integers(0, 4, P)
        q = np.empty(P, np.float32)
        for p in range(P):
            lo, hi = start[p], start[p] + counts[p]
            k = rng.integers(lo, hi)
            right_dead = p + 1 < P and cur[p + 1] < 0
            if kind[p] == 0 or right_dead:
                q[p] = np.float32(0.5) * (depth[k, 0] + depth[k, 1])          # inside interval k
            elif kind[p] == 1:
                q[p] = depth[k, 0] - np.float32(1e-3)                         # just before interval k
            elif kind[p] == 2:
                q[p] = depth[hi - 1, 1] + np.float32(10.0)                    # past every exit: nothing up to the neighbour
            else:
                q[p] = depth[k, 1]                                            # exactly on an exit (<= is inclusive)
        out = ref_lib.find_depth_bound(q, cur, depth)
        np.savez_compressed(os.path.join(OUT, "depth_bound_ref_%s.npz" % name), query=q, curr=cur, depth=depth, out=out)


def _reference_bodies():
    """the helpers of tests/test_reference_modules.py that compile single functions out of the reference files"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "kaolin-wisp_amd"))
    import test_reference_modules as T
    return T


def spc_builder_vectors():
    """spc_builders_ref.npz: the reference's pointcloud_to_octree / dilate_points function bodies (ops/spc/conversions.py:15-48,
    processing.py:13-47) run over the oracle's restatement of the Kaolin leaves: octrees for 0 / 1 / 2 dilation rounds, per-cell
    attribute means, and the dilation of single cells (interior, corner) - the reference's list yields 23 neighbours, not 26."""
    import torch
    T = _reference_bodies()
    torch.Tensor.cuda = lambda self, *a, **k: self                                  # the reference moves its inputs to the GPU
    glb = dict(torch=torch, np=np, spc_ops=T._kaolin_spc_leaves())
    dilate = glb["dilate_points"] = T._reference_function("ops/spc/processing.py", "dilate_points", glb)
    to_octree = T._reference_function("ops/spc/conversions.py", "pointcloud_to_octree", glb)
    rng = np.random.default_rng(2024)
    cloud = rng.uniform(-1, 1, (600, 3)).astype(np.float32)
    cloud[:80] = cloud[80:160]
    cloud[0] = (1.0, -1.0, 1.0)
    att = rng.normal(size=(600, 3)).astype(np.float32)
    out = dict(cloud=cloud, attributes=att, cases=np.array([[4, 0], [4, 1], [3, 2], [5, 1]]))
    for level, rounds in out["cases"]:
        out[f"octree_l{level}_d{rounds}"] = to_octree(torch.from_numpy(cloud), int(level), dilate=int(rounds)).numpy()
    tree, mean = to_octree(torch.from_numpy(cloud), 5, attributes=torch.from_numpy(att))
    out["att_octree_l5"], out["att_mean_l5"] = tree.numpy(), mean.numpy()
    cells = np.array([[9, 9, 9], [0, 0, 0], [31, 0, 17]], dtype=np.int16)
    out["cells"] = cells
    for i, c in enumerate(cells):
        out[f"dilated_{i}"] = dilate(torch.from_numpy(c[None]), 5).numpy()
    np.savez_compressed(os.path.join(OUT, "spc_builders_ref.npz"), **out)


def raygen_vectors():
    """raygen_ref.npz: the reference's generate_pinhole_rays / generate_ortho_rays function bodies (ops/raygen/raygen.py:40-119) on a
    40 x 24 look-at camera with an off-centre principal point (Kaolin's inv_transform_rays restated as R^T (x - t))."""
    import types
    import torch
    T = _reference_bodies()
    from wisp.core import Rays
    from wisp.ops.raygen import LookAtCamera
    W, H = 40, 24
    cam = LookAtCamera(eye=(1.5, 0.8, 2.5), at=(0.1, -0.2, 0.0), up=(0, 1, 0), fov=0.6911112, width=W, height=H, near=0.5, far=7.0,
                       x0=1.75, y0=-0.6, fov_distance=1.3)
    m = cam.view_matrix()[0]
    R, t = m[:3, :3], m[:3, 3]

    class Extrinsics:
        @staticmethod
        def inv_transform_rays(orig, dirs):
            return ((orig - t) @ R)[None], (dirs @ R)[None]

    kcam = types.SimpleNamespace(device=torch.device('cpu'), dtype=torch.float32, width=W, height=H, x0=cam.x0, y0=cam.y0,
                                 near=cam.near, far=cam.far, fov_distance=cam.fov_distance, extrinsics=Extrinsics,
                                 tan_half_fov=lambda axis: cam.tan_half_fov(axis))
    glb = dict(torch=torch, Rays=Rays, CameraFOV=types.SimpleNamespace(HORIZONTAL='horizontal', VERTICAL='vertical'), Camera=object)
    glb["generate_default_grid"] = T._reference_function("ops/raygen/raygen.py", "generate_default_grid", glb)
    glb["_to_ndc_coords"] = T._reference_function("ops/raygen/raygen.py", "_to_ndc_coords", glb)
    grid = T._reference_function("ops/raygen/raygen.py", "generate_centered_pixel_coords", glb)
    py, px = grid(W, H, W, H)
    out = dict(width=W, height=H, x0=cam.x0, y0=cam.y0, fov_distance=cam.fov_distance, tan_h=cam.tan_half_fov('horizontal'),
               tan_v=cam.tan_half_fov('vertical'), rotation=R.numpy(), translation=t.numpy(), pixel_y=py.numpy(), pixel_x=px.numpy(),
               eye=np.asarray(cam.eye, np.float64), at=np.asarray(cam.at, np.float64), fov=cam.fov)
    for name in ("pinhole", "ortho"):
        rays = T._reference_function("ops/raygen/raygen.py", f"generate_{name}_rays", glb)(kcam, (py, px))
        out[f"{name}_origins"], out[f"{name}_dirs"] = rays.origins.numpy(), rays.dirs.numpy()
    np.savez_compressed(os.path.join(OUT, "raygen_ref.npz"), **out)


def spc_vectors():
    out = {}
    # sparse 3-point tree at level 2: points (0,0,0), (3,3,3), (2,1,0)
    pts = np.array([[0, 0, 0], [3, 3, 3], [2, 1, 0]])
    oc = spc.points_to_octree(pts, 2)
    points, pyr, ex = spc.octree_to_spc(oc)
    # hand check: root byte has children 0 (000), 7 (111) and 4 (x=1,y=0,z=0 -> 100b) set = 1 + 128 + 16 = 145
    assert oc[0] == 145 and len(oc) == 4, oc
    # level-1 nodes in morton order: child 0, child 4, child 7.  (0,0,0)->child 0 of node(0,0,0): byte 1;
    # (2,1,0): parent (1,0,0), local (0,1,0) -> slot 2 -> byte 4; (3,3,3): parent (1,1,1), local (1,1,1) -> slot 7 -> byte 128
    assert list(oc[1:]) == [1, 4, 128], oc
    assert pyr.tolist() == [[1, 3, 3, 0], [0, 1, 4, 7]]
    q = np.array([[-0.9, -0.9, -0.9], [0.9, 0.9, 0.9], [0.1, -0.4, -0.9], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1.01, 0, 0]],
                 dtype=np.float32)
    qa = spc.query(oc, ex, q, 2, with_parents=True)
    assert qa[:, 2].tolist() == [4, 6, 5, -1, 6, -1], qa
    rng = np.random.default_rng(3)
    o = rng.normal(size=(64, 3)).astype(np.float32)
    o = (3 * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    d = (rng.uniform(-0.6, 0.6, (64, 3)) - o).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rt = spc.raytrace(oc, points, pyr, ex, o, d, 2, with_exit=True)
    out.update(sp_octree=oc, sp_points=points, sp_pyramid=pyr, sp_exsum=ex, sp_q=q, sp_q_pidx=qa,
               rt_o=o, rt_d=d, rt_ridx=rt[0], rt_pidx=rt[1], rt_depth=rt[2])
    # dense level-2 tree, one axis-aligned ray through the row y=z=-0.75 (cells (k,0,0), k=0..3), origin outside
    ocd = spc.create_dense_octree(2)
    pd, pyd, exd = spc.octree_to_spc(ocd)
    o1 = np.array([[-2.0, -0.75, -0.75]], dtype=np.float32)
    d1 = np.array([[1.0, 0.0, 0.0]], dtype=np.float32)
    r1 = spc.raytrace(ocd, pd, pyd, exd, o1, d1, 2, with_exit=True)
    cells = pd[r1[1]].tolist()
    assert cells == [[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0]], cells
    assert np.allclose(r1[2], [[1.0, 1.5], [1.5, 2.0], [2.0, 2.5], [2.5, 3.0]])
    out.update(dn_octree=ocd, dn_points=pd, dn_pyramid=pyd, dn_exsum=exd, dn_o=o1, dn_d=d1, dn_pidx=r1[1], dn_depth=r1[2])
    np.savez_compressed(os.path.join(OUT, "spc_kats.npz"), **out)


def cell_vectors():
    sys.path.insert(0, os.path.dirname(OUT))
    import adversarial as adv
    s = adv.structured_scalars()
    pts = np.stack([s, s, s], axis=1)
    pos_all, x_some, x_levels = [], [], [0, 8, 13, 15]
    for l, res in enumerate(adv.NGP_RES):
        x, pos = ref_lib.cells_3d(pts, res)
        assert (pos[:, 0] == pos[:, 1]).all() and (pos[:, 0] == pos[:, 2]).all()
        pos_all.append(pos[:, 0].astype(np.int16))
        if l in x_levels:
            x_some.append(x[:, 0])
    np.savez_compressed(os.path.join(OUT, "hashgrid_cells_ref.npz"), scalars=s, res=np.asarray(adv.NGP_RES, dtype=np.int32),
                        pos=np.stack(pos_all), x_levels=np.asarray(x_levels, dtype=np.int32), x=np.stack(x_some))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gradcoords":          # only the vectors added in round 5 (the others stay byte-identical)
        gradcoords_vectors()
        sys.exit(0)
    hashgrid_vectors()
    gradcoords_vectors()
    cell_vectors()
    query_vectors()
    uniform_vectors()
    depth_bound_vectors()
    spc_builder_vectors()
    raygen_vectors()
    spc_vectors()
    print("golden vectors written to", OUT)
