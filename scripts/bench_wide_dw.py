"""Hidden-128 decoder backward (chain kernel + dW kernel) at 2 M samples: time and parameter gradients of ONE library variant
(WISP_WIDE_DW=1: the barrier-per-stage dW kernel of rounds 2-4; default: the producer / consumer pipeline of round 5).
usage: python scripts/bench_wide_dw.py OUT.pt [S,S,...]     ->  prints the medians, saves grad_params / grad_feats for a bitwise comparison"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch
import wisp._C as C
dev = "cuda:0"
out = sys.argv[1]
res = {}
sizes = tuple(int(v) for v in sys.argv[2].split(',')) if len(sys.argv) > 2 else (2_000_000, 70_001)
for S in sizes:
    g = torch.Generator(device=dev).manual_seed(0)
    d = torch.nn.functional.normalize(torch.randn(S, 3, device=dev, generator=g), dim=1)
    g_rgb = torch.randn(S, 3, device=dev, generator=g) * 1e-3
    g_den = torch.randn(S, 1, device=dev, generator=g) * 1e-3
    feats = (torch.randn(S, 32, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    n = int(C.lib.wisp_nerf_mlp_param_count(32, 128, 4))
    params = torch.randn(n, device=dev, generator=g) * 0.1
    gp = torch.zeros_like(params)

    def run():
        gp.zero_()
        return C.nerf_mlp_backward(feats, d, params, g_rgb, g_den, 32, 128, 4, True, grad_params=gp)

    for _ in range(3):
        gf, _ = run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gf, _ = run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    print(f"WISP_WIDE_DW={os.environ.get('WISP_WIDE_DW', '')!r} S={S}: backward median {sorted(ts)[len(ts) // 2]:8.1f} us  (min {min(ts):8.1f})")
    res[S] = (gp.cpu().clone(), gf.float().cpu().clone())
torch.save(res, out)
