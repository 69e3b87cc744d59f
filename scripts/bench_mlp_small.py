"""Decoder forward / backward (hidden 64, bf16, per-ray view codes) against the sample count: where the fixed cost of a launch ends."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch
import wisp._C as C
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
n = int(C.lib.wisp_nerf_mlp_param_count(32, 64, 4))
params = torch.randn(n, device=dev, generator=g) * 0.1
gp = torch.zeros_like(params)
for S in (1024, 8192, 32768, 65536, 131072, 262144, 524288, 1 << 20, 1 << 21):
    R = max(64, S // 54)
    feats = (torch.randn(S, 32, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=1)
    ridx = torch.sort(torch.randint(0, R, (S,), device=dev, generator=g)).values
    code = C.nerf_mlp_dir_code(dirs, 4)
    g_rgb = torch.randn(S, 3, device=dev, generator=g) * 1e-3
    g_den = torch.randn(S, 1, device=dev, generator=g) * 1e-3
    def fwd(): return C.nerf_mlp_forward(feats, None, params, 32, 64, 4, True, ray_code=(ridx, code))
    def bwd(): return C.nerf_mlp_backward(feats, None, params, g_rgb, g_den, 32, 64, 4, True, grad_params=gp, ray_code=(ridx, code))
    out = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        out[name] = sorted(ts)[10]
    print(f"S {S:>8}: fwd {out['fwd']:7.1f} us  bwd (kernel + reduce) {out['bwd']:7.1f} us   per 2^18 samples: fwd {out['fwd'] * 262144 / S:7.1f}  bwd {out['bwd'] * 262144 / S:7.1f}")
