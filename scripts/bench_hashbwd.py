"""hashgrid backward microbench at the training shape (ray-ordered coordinates, 2^21 samples, bf16 gradients): time per launch and
scratch size; run under different WISP_HG_* switches to A/B the bin geometry."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaolin-wisp_amd"))
import wisp._C as C
from wisp.models.grids import HashGrid
from wisp.accelstructs import OctreeAS

dev = torch.device("cuda:0")
S = 1 << 21
torch.manual_seed(0)
R = S // 40
o = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=1) * 3.2
tgt = (torch.rand(R, 3, device=dev) - 0.5)
d = torch.nn.functional.normalize(tgt - o, dim=1)
t = 2.4 + torch.rand(R, 1, device=dev) * 1.4 + torch.arange(40, device=dev).float()[None, :] * (4.0 / 2048)
coords = (o[:, None, :] + d[:, None, :] * t[..., None]).reshape(-1, 3).clamp(-1, 1).contiguous()
blas = OctreeAS.make_dense(level=2)
grid = HashGrid.from_geometric(blas, feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=0.1, codebook_bitwidth=19,
                               min_grid_res=16, max_grid_res=512).to(dev)
cb = grid.codebook
res = [int(r) for r in cb.resolutions.reshape(-1).tolist()]
g = (torch.randn(S, 32, device=dev) * 1e-3).to(torch.bfloat16)
out = torch.zeros_like(cb.feats.detach())


def run():
    C.hashgrid_interpolate_backward(coords, g, tuple(cb.feats.shape), cb.begin_idxes, res, 19, 30, out=out)


for _ in range(40):           # the slot fit settles
    run()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
fits = [f.last for f in getattr(C, "_slot_fits", {}).values() if f.last]
ws = max((f["workspace_bytes"] for f in fits), default=0)
print("WISP_HG_CHUNK_FLOATS=%s: hashgrid_bwd %.1f us (median of 20), scratch %.2f GB, checksum %.6e" %
      (os.environ.get("WISP_HG_CHUNK_FLOATS", "-"), sorted(ts)[10], ws / 2 ** 30, float(out.double().abs().sum())))
