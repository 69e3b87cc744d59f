"""cProfile of the unchanged trainer's iterate() on the GPU box: where the HOST time of the drop-in regime goes (the GPU is busy a third
of the step).  Prints the top functions by own time and by cumulative time for 300 steady-state iterations (no prune inside)."""
import cProfile, os, pstats, sys, io, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.environ.get("WISP_PKG_DIR") or os.path.join(ROOT, "kaolin-wisp_amd"))
import torch, numpy as np
import bench, synlego
from wisp.datasets import MultiviewTensorDataset, SampleRays
from wisp.trainers import MultiviewTrainer, ConfigMultiviewTrainer, ConfigAdamW

dev = torch.device("cuda:0")
cells = synlego.occupied_cells(7, device=dev)
pipe = bench.build_pipeline(dev, 64, 2048, cells)
o, d, rgb = synlego.ray_bank(2 ** 21, seed=1, device=dev)
ds = MultiviewTensorDataset(o.view(8, -1, 3), d.view(8, -1, 3), rgb.view(8, -1, 3), synlego.NEAR, synlego.FAR, transform=SampleRays(4096))
cfg = ConfigMultiviewTrainer(optimizer=ConfigAdamW(lr=1e-3, eps=1e-16, weight_decay=1e-6), grid_lr_weight=500.0, enable_amp=True,
                             scheduler=True, prune_every=-1, rgb_loss_type='huber', rgb_loss_denom='rays', max_epochs=10 ** 6,
                             target_sample_size=2 ** 18)
tr = MultiviewTrainer(cfg, pipe, ds, device=dev)
tr.is_optimization_running = True
for _ in range(60):
    tr.iterate()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    tr.iterate()
torch.cuda.synchronize()
print(f"unprofiled: {1e3 * (time.perf_counter() - t0) / N:.3f} ms per iterate, rays {ds.transform.num_samples}, samples {pipe.tracer.get_prev_num_samples()}")
if os.environ.get('NO_PROFILE'):
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.iterate()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(38)
    lines = s.getvalue().splitlines()
    print(f"==== by {key} (per-call figures are over {N} iterations)")
    print("\n".join(l[:170] for l in lines[4:50]))
