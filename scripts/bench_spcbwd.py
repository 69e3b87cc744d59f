"""Microbenchmark of the order-free trilinear / codebook backward (csrc/spc_grad.hip) on the VQAD bench's sample distribution:
SynLego V8 point cloud -> level-8 octree, 4 LODs (levels 5-8), 'voxel' march with 16 samples per cell, ~2 M samples.
Prints the median time of the whole call; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
    MARCH=ray python scripts/bench_spcbwd.py      # unaligned runs: the general segmented scan
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch, numpy as np
import synlego, wisp._C as C
from wisp.accelstructs import OctreeAS
from wisp.core import Rays
from wisp.models.grids import CodebookOctreeGrid, OctreeGrid

dev = "cuda:0"
torch.manual_seed(0)
march = os.environ.get("MARCH", "voxel")
cloud = synlego.v8_pointcloud(1 << 21, res=400, device=dev)
blas = OctreeAS.from_pointcloud(cloud, 8)
grid = CodebookOctreeGrid(blas, feature_dim=5, num_lods=4, multiscale_type='sum', feature_std=0.01, codebook_bitwidth=4).to(dev)
ogrid = OctreeGrid(blas, feature_dim=5, num_lods=4, multiscale_type='sum', feature_std=0.01).to(dev)
o, d, _ = synlego.ray_bank(1 << 18, res=400, seed=1000, device=dev, with_gt=False)
rays = Rays(o, d, dist_min=synlego.NEAR, dist_max=synlego.FAR)
steps = 16 if march == "voxel" else 512
rm = grid.raymarch(rays, level=grid.active_lods[-1], num_samples=steps, raymarch_type=march)
target = int(os.environ.get("SAMPLES", 2_000_000))
S = min(rm.samples.shape[0], target) // 64 * 64
samples = rm.samples[:S].contiguous()
L, F = 4, 5
levels = grid.active_lods[:L]
chain = blas.query_chain(samples, levels[-1], grid.base_lod)
trk = grid.trinkets.int().to(dev)
g = torch.randn(S, F, device=dev) * 1e-3
logits = [f.detach() for f in grid.features[:L]]
dicts = [t.detach() for t in grid.dictionary[:L]]
gl = [torch.zeros_like(t) for t in logits]
gdc = [torch.zeros_like(t) for t in dicts]
gf = [torch.zeros_like(t) for t in ogrid.features[:L]]
shapes = [tuple(t.shape) for t in ogrid.features[:L]]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


cb = lambda: C.codebook_trilinear_multi_backward(samples, chain, blas.points, trk, logits, dicts, g, levels, True, out=(gl, gdc))
oc = lambda: C.spc_trilinear_multi_backward(samples, chain, blas.points, trk, g, shapes, levels, True, out=gf)
runs = (chain[1:, L - 1] != chain[:-1, L - 1]).sum().item() + 1
print(f"march {march}: S = {S}, rows per level {[t.shape[0] for t in logits]}, finest-level runs {runs} ({S / runs:.1f} samples each)")
t_cb, t_oc = timeit(cb), timeit(oc)
chk = sum(float(t.double().abs().sum()) for t in gl) / 23, sum(float(t.double().abs().sum()) for t in gf) / 23
print(f"codebook multi bwd {t_cb:8.1f} us   octree multi bwd {t_oc:8.1f} us   checksums {chk[0]:.9e} {chk[1]:.9e}")
