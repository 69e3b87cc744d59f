"""Experiment harness: time the hash-grid backward alone on realistic samples (SynLego raymarch) with HIP events,
for the env configuration it is started under.  Prints one line."""
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
rm = blas.raymarch(Rays(o, d, dist_min=1.0, dist_max=5.0), 'ray', 2048)
coords = rm.samples
S = coords.shape[0]
res = [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512]
sizes = [min(2 ** 19, r ** 3) for r in res]
begin = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64, device=dev)
g = torch.randn(S, 32, device=dev).bfloat16()
grad = torch.zeros(int(begin[-1]), 2, device=dev)
for _ in range(3):
    C.hashgrid_interpolate_backward(coords, g, grad.shape, begin, res, 19, zero_from_col=30, out=grad)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    C.hashgrid_interpolate_backward(coords, g, grad.shape, begin, res, 19, zero_from_col=30, out=grad)
e1.record(); torch.cuda.synchronize()
print(f"S={S} env=[{' '.join(k+'='+v for k,v in os.environ.items() if k.startswith('WISP_'))}] bwd_ms={e0.elapsed_time(e1)/10:.3f}")
