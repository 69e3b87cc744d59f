"""rocprofv3 target: a few forward / backward launches of the hidden-128 decoder at ~2 M samples."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch
import wisp._C as C
dev = "cuda:0"
S, H = 2_000_000, 128
n = int(C.lib.wisp_nerf_mlp_param_count(32, H, 4))
params = torch.randn(n, device=dev) * 0.1
feats = torch.randn(S, 32, device=dev).bfloat16()
dirs = torch.nn.functional.normalize(torch.randn(S, 3, device=dev), dim=1)
gr = torch.randn(S, 3, device=dev); gd = torch.randn(S, 1, device=dev)
for _ in range(4):
    C.nerf_mlp_forward(feats, dirs, params, 32, H, 4, True)
    C.nerf_mlp_backward(feats, dirs, params, gr, gd, 32, H, 4, True)
torch.cuda.synchronize()
# plain event timing of the pair (no profiler needed)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
for _ in range(10):
    C.nerf_mlp_forward(feats, dirs, params, 32, H, 4, True)
e[1].record()
for _ in range(10):
    C.nerf_mlp_backward(feats, dirs, params, gr, gd, 32, H, 4, True)
e[2].record()
torch.cuda.synchronize()
print("hidden %d, %d samples: forward %.3f ms, backward %.3f ms" % (H, S, e[0].elapsed_time(e[1]) / 10, e[1].elapsed_time(e[2]) / 10))
