"""Does the hash-grid forward (waits for L2-miss fills) overlap with the decoder forward (matrix pipe + vector ALU) when the two run
on two streams, chunk i + 1 of the lookup beside chunk i of the decoder?  Same coordinates / tables / decoder as the training step
at 2 M samples; serial issue against K-chunk pipelines.  (Round 3 tried whole kernels side by side - the occupancy count beside
the step's kernels - and found nothing to gain; this pair has complementary bounds.)  Also the backward pair: decoder backward
chunk i + 1 beside the hash-grid backward's emit... not possible (one reduce over all records), so forward only."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaolin-wisp_amd"))
import wisp._C as C
from wisp.models.grids import HashGrid
from wisp.accelstructs import OctreeAS

dev = torch.device("cuda:0")
S = 1 << 21
torch.manual_seed(0)
R = S // 50
o = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=1) * 3.2
d = torch.nn.functional.normalize((torch.rand(R, 3, device=dev) - 0.5) - o, dim=1)
t = 2.4 + torch.rand(R, 1, device=dev) * 1.4 + torch.arange(50, device=dev).float()[None, :] * (4.0 / 2048)
coords = (o[:, None, :] + d[:, None, :] * t[..., None]).reshape(-1, 3).clamp(-1, 1).contiguous()
ridx = torch.arange(R, device=dev).repeat_interleave(50).contiguous()
grid = HashGrid.from_geometric(OctreeAS.make_dense(level=2), feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=0.1,
                               codebook_bitwidth=19, min_grid_res=16, max_grid_res=512).to(dev)
cb = grid.codebook
table = cb.feats.detach().to(torch.bfloat16)
res = [int(r) for r in cb.resolutions.reshape(-1).tolist()]
n = int(C.lib.wisp_nerf_mlp_param_count(32, 64, 4))
params = torch.randn(n, device=dev) * 0.2
code = C.nerf_mlp_dir_code(d)
side = torch.cuda.Stream()


def serial():
    f = C.hashgrid_interpolate(coords, table, cb.begin_idxes, res, 19, 30)
    return C.nerf_mlp_forward(f, None, params, 32, 64, 4, True, ray_code=(ridx, code))


def pipelined(K):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    bounds = [(S * k // K) // 64 * 64 for k in range(K)] + [S]
    evs = []
    outs = []
    for k in range(K):
        a, b = bounds[k], bounds[k + 1]
        f = C.hashgrid_interpolate(coords[a:b], table, cb.begin_idxes, res, 19, 30)          # main stream
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            outs.append(C.nerf_mlp_forward(f, None, params, 32, 64, 4, True, ray_code=(ridx[a:b], code)))
            f.record_stream(side)
    main.wait_stream(side)
    return outs


def timeit(fn, reps=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


ref = serial()
got = pipelined(4)
assert torch.equal(torch.cat([g[0] for g in got]), ref[0]) and torch.equal(torch.cat([g[1] for g in got]), ref[1])
print(f"serial: lookup + decoder forward {timeit(serial):7.1f} us")
print(f"lookup alone {timeit(lambda: C.hashgrid_interpolate(coords, table, cb.begin_idxes, res, 19, 30)):7.1f} us")
for K in (2, 3, 4, 6, 8):
    print(f"two streams, {K} chunks: {timeit(lambda: pipelined(K)):7.1f} us")
