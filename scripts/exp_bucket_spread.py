"""How evenly do the gradient records of one training step fall into (bucket, emitting workgroup) slots of the binned hash-grid
backward - with the bucket = 8192 CONSECUTIVE table rows (what csrc/hashgrid.hip does) against buckets made of interleaved strips
of 2^k rows (bucket = (row >> k) % buckets)?  Takes the coordinates of a real step of the bench run (learned occupancy), emulates
the run merge (one tail per run of samples in the same cell inside a 64-sample group) and histograms on the GPU."""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "kaolin-wisp_amd"))
import torch
import wisp._C as C
import bench

seen = {}
orig = C.hashgrid_interpolate_backward
def spy(coords, grad_feats, codebook_shape, first_idx, resolutions, codebook_bitwidth, *a, **k):
    if coords.shape[0] > (1 << 20):
        seen["coords"] = coords.detach().clone()
        seen["res"] = [int(r) for r in resolutions]
        seen["bw"] = int(codebook_bitwidth)
    return orig(coords, grad_feats, codebook_shape, first_idx, resolutions, codebook_bitwidth, *a, **k)
C.hashgrid_interpolate_backward = spy
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-pmc", "--no-configs", "--dropin-steps", "0"])
coords, res, bw = seen["coords"], seen["res"], seen["bw"]
n = coords.shape[0]
dev = coords.device
pieces = (n + 1023) // 1024
WGS = int(os.environ.get("EXP_WGS", "256"))
per_wg = (pieces + WGS - 1) // WGS
ntiles = (pieces + per_wg - 1) // per_wg
i = torch.arange(n, device=dev)
wg = (i // 1024) % ntiles
print(f"n = {n}, pieces {pieces}, emitting workgroups {ntiles}")
T = 1 << bw
for l in range(15):
    r = res[l]
    cell, frac, corners = C.hashgrid_cells(coords, r, bw)
    same = (cell[1:] == cell[:-1]).all(dim=1) & ((i[1:] % 64) != 0)
    tail = torch.ones(n, dtype=torch.bool, device=dev)
    tail[:-1] = ~same
    entries = min(r ** 3, T)
    chunks = (entries + 8191) // 8192
    idx = corners[tail].long().reshape(-1)
    w = wg[tail].repeat_interleave(8)
    ok = idx < entries
    idx, w = idx[ok], w[ok]
    total = idx.numel()
    line = f"level {l:2d} res {r:4d} {'dense ' if r ** 3 < T else 'hashed'} buckets {chunks:3d} records {total:8d} mean/slot {total / (chunks * ntiles):7.1f} | fullest/mean:"
    def spread(b):
        h = torch.bincount(b * ntiles + w, minlength=chunks * ntiles)
        per_bucket = h.view(chunks, ntiles).sum(1)
        return h.max().item() / (total / (chunks * ntiles)), per_bucket.max().item() / per_bucket.float().mean().item(), int(h.max())
    a = spread(idx >> 13)
    pw = torch.bincount(w, minlength=ntiles).float()
    hh = torch.bincount((idx >> 13) * ntiles + w, minlength=chunks * ntiles).view(chunks, ntiles).float()
    rel = hh / (pw[None, :] / chunks).clamp(min=1)
    line += f" [per-workgroup totals max/mean {pw.max().item() / pw.mean().item():4.2f}, min/mean {pw.min().item() / pw.mean().item():4.2f}; slot / (its workgroup's mean slot) max {rel.max().item():4.2f}]"
    line += f" consecutive {a[0]:5.2f} (bucket load max/mean {a[1]:4.2f}, fullest {a[2]})"
    if r ** 3 < T and chunks > 1:
        for k in (4, 5, 6):
            st = idx >> k
            q, rr = st // chunks, st % chunks
            m = (((q * 40503) & 0xffff) * chunks) >> 16
            for name, b in (("plain", rr), ("rot", (rr + q) % chunks), ("mix", (rr + m) % chunks)):
                sp = spread(b)
                line += f"; {1 << k}/{name}: {sp[0]:5.2f} ({sp[1]:4.2f}, {sp[2]})"
    print(line)
