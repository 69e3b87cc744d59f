"""Host-side profile of the training step at the reference's default batch (2^18 samples): where does Python time go?"""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch
import bench, synlego
from wisp.core import Rays
from wisp.trainers import MultiviewTrainStep
dev = torch.device("cuda", 0)
cells = synlego.occupied_cells(7, device=dev)
pipe = bench.build_pipeline(dev, 64, 2048, cells)
tr = MultiviewTrainStep(pipe, prune_every=-1, target_sample_size=2 ** 18, enable_amp=True)
o, d, rgb = synlego.ray_bank(2 ** 18, seed=1, device=dev)
R = 6200
def step():
    idx = torch.randint(0, o.shape[0], (R,), device=dev)
    tr.step(Rays(o.index_select(0, idx), d.index_select(0, idx), dist_min=1.0, dist_max=5.0), rgb.index_select(0, idx))
for _ in range(10): step()
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t) * 10)
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:6500])
