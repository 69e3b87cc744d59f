"""Per-iteration wall times of the unchanged-trainer regime with the prune inside (what bench.py's dropin_regime times)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaolin-wisp_amd"))
import torch, numpy as np
import bench, synlego
from wisp.accelstructs import OctreeAS
from wisp.datasets import MultiviewTensorDataset, SampleRays
from wisp.trainers import MultiviewTrainer, ConfigMultiviewTrainer, ConfigAdamW

dev = torch.device("cuda:0")
cells = OctreeAS.make_dense(level=7).points[-(128 ** 3):].to(dev) if os.environ.get("DENSE", "1") == "1" else synlego.occupied_cells(7, device=dev)
pipe = bench.build_pipeline(dev, 64, 2048, cells)
o, d, rgb = synlego.ray_bank(2 ** 21, seed=1, device=dev)
ds = MultiviewTensorDataset(o.view(8, -1, 3), d.view(8, -1, 3), rgb.view(8, -1, 3), synlego.NEAR, synlego.FAR, transform=SampleRays(4096))
cfg = ConfigMultiviewTrainer(optimizer=ConfigAdamW(lr=1e-3, eps=1e-16, weight_decay=1e-6), grid_lr_weight=500.0, enable_amp=True,
                             scheduler=True, prune_every=100, rgb_loss_type='huber', rgb_loss_denom='rays', max_epochs=10 ** 6,
                             target_sample_size=2 ** 18)
tr = MultiviewTrainer(cfg, pipe, ds, device=dev)
tr.is_optimization_running = True
times = []
for it in range(460):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.iterate()
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
t = np.array(times)
for a in range(0, 460, 20):
    seg = t[a:a + 20]
    print(f"iters {a:3d}-{a + 19:3d}: mean {seg.mean():7.3f} ms  max {seg.max():8.3f} ms (at {a + int(seg.argmax())})  rays {ds.transform.num_samples} samples {pipe.tracer.get_prev_num_samples()}")
print("cells", int(pipe.nef.grid.blas.pyramid[0, 7]))
