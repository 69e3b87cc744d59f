"""How many records of the binned hash-grid backward repeat the table entry of the record right before them in the same slot?
(Consecutive cells of a ray share a face: four of a tail's eight entries reappear in the next tail, and equal entries land in the
same bucket, i.e. next to each other in that slot.)  Decodes the scratch of one launch at the training shape; WISP_HG_SLOT_FIT=0."""
import os, sys
os.environ["WISP_HG_SLOT_FIT"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaolin-wisp_amd"))
import torch
import wisp._C as C
from wisp.models.grids import HashGrid
from wisp.accelstructs import OctreeAS

dev = torch.device("cuda:0")
S = 1 << 21
torch.manual_seed(0)
R = S // 40
o = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=1) * 3.2
d = torch.nn.functional.normalize((torch.rand(R, 3, device=dev) - 0.5) - o, dim=1)
t = 2.4 + torch.rand(R, 1, device=dev) * 1.4 + torch.arange(40, device=dev).float()[None, :] * (4.0 / 2048)
coords = (o[:, None, :] + d[:, None, :] * t[..., None]).reshape(-1, 3).clamp(-1, 1).contiguous()
grid = HashGrid.from_geometric(OctreeAS.make_dense(level=2), feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=0.1,
                               codebook_bitwidth=19, min_grid_res=16, max_grid_res=512).to(dev)
cb = grid.codebook
res = [int(r) for r in cb.resolutions.reshape(-1).tolist()]
g = (torch.randn(S, 32, device=dev) * 1e-3).to(torch.bfloat16)
out = torch.zeros_like(cb.feats.detach())
C.hashgrid_interpolate_backward(coords, g, tuple(cb.feats.shape), cb.begin_idxes, res, 19, 30, out=out)
torch.cuda.synchronize()
ws = next(iter(C._bwd_ws.values()))
# bin_plan (csrc/hashgrid.hip) for the queue emitter at this size: 512 emitting workgroups of four 1024-sample pieces
ntiles, tile_samples = 512, 4096
chunks, caps = [], []
for l in range(15):
    dense = res[l] ** 3 <= 2 ** 19
    entries = min(res[l] ** 3, 2 ** 19)
    ch = (entries + 8191) // 8192
    cap = (tile_samples * 8 + ch - 1) // ch
    cap = cap * 2 if dense else cap + cap // 4
    cap = max(cap, 128)
    if dense:
        cap = min(cap, 2048 * 4)
    cap = min(cap, tile_samples * 8)
    cap = (((cap + 31) // 32) | 1) * 32
    chunks.append(ch); caps.append(cap)
cnt_cells = sum(c * ntiles for c in chunks) + ntiles
count_bytes = (cnt_cells * 4 + 255) // 256 * 256
counts = ws[:count_bytes].view(torch.int32)
recs = ws[count_bytes:].view(torch.int32)
cnt_base = rec_base = 0
for l in range(15):
    ch, cap = chunks[l], caps[l]
    c = counts[cnt_base:cnt_base + ch * ntiles].view(ch, ntiles).long()
    if l >= 7:
        r = recs[rec_base * 2:(rec_base + ch * ntiles * cap) * 2].view(ch, ntiles, cap, 2)
        loc = (r[..., 0] & 0x7f) | ((r[..., 1] & 0x7f) << 7)
        pos = torch.arange(cap, device=dev)[None, None, :]
        live = pos < c[..., None]
        same1 = (loc[..., 1:] == loc[..., :-1]) & live[..., 1:]
        same2 = (loc[..., 2:] == loc[..., :-2]) & live[..., 2:]
        # duplicates inside the 64-record chunk a wave handles at once (any distance)
        n_live = int(live.sum())
        lc = loc.clone(); lc[~live] = -1
        pad = (-cap) % 64
        if pad:
            lc = torch.nn.functional.pad(lc, (0, pad), value=-1)
        ck = lc.view(ch, ntiles, -1, 64)
        srt, _ = ck.sort(dim=-1)
        dup_any = int(((srt[..., 1:] == srt[..., :-1]) & (srt[..., 1:] >= 0)).sum())
        print(f"level {l:2d} res {res[l]:3d}: {n_live / S:5.2f} records/sample, fullest slot {int(c.max())}/{cap}; "
              f"same entry as the record before {int(same1.sum()) / n_live:.3f}, as the one two before {int(same2.sum()) / n_live:.3f}, "
              f"any earlier record of its 64-record chunk {dup_any / n_live:.3f}")
    cnt_base += ch * ntiles
    rec_base += ch * ntiles * cap
