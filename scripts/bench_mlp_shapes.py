"""Decoder forward / backward time by feature-row shape (2 M samples, bf16 compute, per-ray view codes): which of 'narrow' and
'fp32 I/O' costs the octree / codebook fields' decoder launches their 14 % against the hash-grid field's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch, numpy as np
import wisp._C as C
dev = "cuda:0"
S, R = 2_000_000, 40_000
g = torch.Generator(device=dev).manual_seed(0)
ridx = torch.sort(torch.randint(0, R, (S,), device=dev, generator=g)).values
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=1)
code = C.nerf_mlp_dir_code(d)
g_rgb = torch.randn(S, 3, device=dev, generator=g); g_den = torch.randn(S, 1, device=dev, generator=g)

def timeit(fn, reps=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]

for dt, in_dim in ((torch.bfloat16, 32), (torch.float32, 32), (torch.bfloat16, 8), (torch.float32, 8), (torch.float32, 5), (torch.bfloat16, 5)):
    feats = (torch.randn(S, in_dim, device=dev, generator=g) * 0.5).to(dt)
    n = int(C.lib.wisp_nerf_mlp_param_count(in_dim, 64, 4))
    params = torch.randn(n, device=dev, generator=g) * 0.2
    gp = torch.zeros_like(params)
    f = timeit(lambda: C.nerf_mlp_forward(feats, None, params, in_dim, 64, 4, True, ray_code=(ridx, code)))
    b = timeit(lambda: C.nerf_mlp_backward(feats, None, params, g_rgb, g_den, in_dim, 64, 4, True, grad_params=gp, ray_code=(ridx, code)))
    print(f"{str(dt).split('.')[-1]:9s} in_dim {in_dim:2d}: fwd {f:7.1f} us   bwd {b:7.1f} us")
