"""A/B timing of the hot kernels for the library WISP_HIP_LIB points at (default: the in-tree build), on the bench's sample
distribution (SynLego raymarch, ~2 M samples).  Box-to-box spread is +-5 %, so variants are compared inside ONE gpurun call:
    for lib in csrc/ab/base.so csrc/libwisp_hip.so; do WISP_HIP_LIB=$lib python scripts/ab_kernels.py; done
"""
import os, sys, ctypes
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
coords, ridx = rm.samples, rm.ridx
S = coords.shape[0]
res = [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512]
sizes = [min(2 ** 19, r ** 3) for r in res]
begin = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64, device=dev)
table = (torch.randn(int(begin[-1]), 2, device=dev) * 0.1).bfloat16()
g = torch.randn(S, 32, device=dev).bfloat16()
grad = torch.zeros(int(begin[-1]), 2, device=dev)
n = int(C.lib.wisp_nerf_mlp_param_count(32, 64, 4))
params = torch.randn(n, device=dev) * 0.1
gp = torch.zeros_like(params)
gr = torch.randn(S, 3, device=dev); gd = torch.randn(S, 1, device=dev)
dirs = d.index_select(0, ridx)
rgb = torch.empty(S, 3, device=dev); den = torch.empty(S, 1, device=dev); gf = torch.empty_like(g)
ws = torch.empty(int(C.lib.wisp_nerf_mlp_workspace_floats()), device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
mlp_f = lambda: C.lib.wisp_nerf_mlp_fwd(P(g), 2, P(dirs), S, 32, 64, 4, P(params), 2, P(rgb), P(den), st)
mlp_b = lambda: C.lib.wisp_nerf_mlp_bwd(P(g), 2, P(dirs), S, 32, 64, 4, P(params), 2, P(gr), P(gd), P(gf), P(gp), P(ws), ws.numel() * 4, st)


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


code = C.nerf_mlp_dir_code(d, 4)
mlp_fr = lambda: C.nerf_mlp_forward(g, None, params, 32, 64, 4, True, ray_code=(ridx, code))
mlp_br = lambda: C.nerf_mlp_backward(g, None, params, gr, gd, 32, 64, 4, True, grad_params=gp, ray_code=(ridx, code))
t = dict(mlp_fwd_rays=timeit(mlp_fr), mlp_bwd_rays=timeit(mlp_br),
    hg_fwd=timeit(lambda: C.hashgrid_interpolate(coords, table, begin, res, 19, 30)),
    hg_bwd=timeit(lambda: C.hashgrid_interpolate_backward(coords, g, grad.shape, begin, res, 19, zero_from_col=30, out=grad)),
    mlp_fwd=timeit(mlp_f), mlp_bwd=timeit(mlp_b))
# the reference trainer's batch: 2^18 samples
S2 = 1 << 18
c2, g2 = coords[:S2].contiguous(), g[:S2].contiguous()
t["hg_bwd_2p18"] = timeit(lambda: C.hashgrid_interpolate_backward(c2, g2, grad.shape, begin, res, 19, zero_from_col=30, out=grad))
t["hg_fwd_2p18"] = timeit(lambda: C.hashgrid_interpolate(c2, table, begin, res, 19, 30))

print(f"{os.path.basename(C.LIB_PATH):20s} S={S} " + "  ".join(f"{k} {v:7.1f} us" for k, v in t.items()), flush=True)
