"""Timing of the fused decoder entry points over a sweep of sample counts (HIP events, median of reps)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaolin-wisp_amd"))
import wisp._C as C

dev = torch.device("cuda:0")
n = int(C.lib.wisp_nerf_mlp_param_count(32, 64, 4))
params = torch.randn(n, device=dev) * 0.1
gp = torch.zeros_like(params)


def timeit(fn, reps=15):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for S in [32, 32 * 1024, 262144, 1 << 20, 1 << 21, 1 << 22]:
    feats = torch.randn(S, 32, device=dev).bfloat16(); dirs = torch.nn.functional.normalize(torch.randn(S, 3, device=dev), dim=1)
    gr = torch.randn(S, 3, device=dev); gd = torch.randn(S, 1, device=dev)
    f = timeit(lambda: C.nerf_mlp_forward(feats, dirs, params, 32, 64, 4, True))
    b = timeit(lambda: C.nerf_mlp_backward(feats, dirs, params, gr, gd, 32, 64, 4, True, grad_params=gp))
    print(f"S={S:8d}  fwd {f:8.1f} us   bwd {b:8.1f} us", flush=True)
