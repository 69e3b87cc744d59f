"""Is hashgrid_fwd bound by cache misses or by the scattered-request rate?  Same ray-ordered coordinates, same levels,
tables of 2^19 vs 2^14 entries per hashed level (the small ones are cache resident)."""
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


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


blas = OctreeAS.make_dense(level=2)
for bw in (19, 14):
    grid = HashGrid.from_geometric(blas, feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=0.1,
                                   codebook_bitwidth=bw, min_grid_res=16, max_grid_res=512).to(dev)
    cb = grid.codebook
    table = cb.feats.detach().to(torch.bfloat16)
    res = [int(r) for r in cb.resolutions.reshape(-1).tolist()]
    f = timeit(lambda: C.hashgrid_interpolate(coords, table, cb.begin_idxes, res, bw, None))
    print(f"bitwidth {bw}: table {table.numel() * 2 / 2 ** 20:.1f} MB  fwd {f:.1f} us for {coords.shape[0]} samples", flush=True)

# does the texture-address path merge lanes that ask for the same address?
grid = HashGrid.from_geometric(blas, feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=0.1,
                               codebook_bitwidth=19, min_grid_res=16, max_grid_res=512).to(dev)
cb = grid.codebook
table = cb.feats.detach().to(torch.bfloat16)
res = [int(r) for r in cb.resolutions.reshape(-1).tolist()]
same = coords[:1].expand(S, 3).contiguous()
per_wave = coords[::64][: S // 64].repeat_interleave(64, dim=0).contiguous()
for name, c in (("ray-ordered", coords), ("one point for all", same), ("one point per wave", per_wave)):
    f = timeit(lambda: C.hashgrid_interpolate(c, table, cb.begin_idxes, res, 19, None))
    print(f"{name:20s}: {f:.1f} us", flush=True)
