"""Is the hash-grid forward bound by L2 misses?  Same samples, same levels, tables of 2^19 / 2^17 / 2^15 rows per hashed level
(20.9 MB / 6.4 MB / 2.0 MB in bf16: beyond / about / well inside one XCD's 4 MB L2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch, numpy as np
import synlego, wisp._C as C
from wisp.accelstructs import OctreeAS
from wisp.core import Rays
dev = "cuda:0"
cells = synlego.occupied_cells(7, device=dev)
blas = OctreeAS.from_quantized_points(cells, 7)
o, d, _ = synlego.ray_bank(49623, seed=5, device=dev, with_gt=False)
coords = blas.raymarch(Rays(o, d, dist_min=1.0, dist_max=5.0), 'ray', 2048).samples
S = coords.shape[0]
res = [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]
for bw in (19, 17, 15, 13):
    sizes = [min(2 ** bw, r ** 3) for r in res]
    begin = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64, device=dev)
    table = (torch.randn(int(begin[-1]), 2, device=dev) * 0.1).bfloat16()
    g = torch.randn(S, 32, device=dev).bfloat16()
    grad = torch.zeros(int(begin[-1]), 2, device=dev)
    f = timeit(lambda: C.hashgrid_interpolate(coords, table, begin, res, bw, 30))
    b = timeit(lambda: C.hashgrid_interpolate_backward(coords, g, grad.shape, begin, res, bw, zero_from_col=30, out=grad))
    print(f"T=2^{bw}: table {table.numel() * 2 / 2**20:6.1f} MiB  fwd {f:7.1f} us  bwd {b:7.1f} us  (S={S})", flush=True)
