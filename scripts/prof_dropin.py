"""Where the unchanged-trainer regime spends its step: wall time per iterate(), kernel list (torch.profiler), host gaps."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaolin-wisp_amd"))
import torch
import bench, synlego
from wisp.datasets import MultiviewTensorDataset, SampleRays
from wisp.trainers import MultiviewTrainer, ConfigMultiviewTrainer, ConfigAdamW

dev = torch.device("cuda:0")
cells = synlego.occupied_cells(7, device=dev)
pipe = bench.build_pipeline(dev, 64, 2048, cells)
o, d, rgb = synlego.ray_bank(2 ** 20, seed=1, device=dev)
ds = MultiviewTensorDataset(o.view(8, -1, 3), d.view(8, -1, 3), rgb.view(8, -1, 3), synlego.NEAR, synlego.FAR, transform=SampleRays(4096))
cfg = ConfigMultiviewTrainer(optimizer=ConfigAdamW(lr=1e-3, eps=1e-16, weight_decay=1e-6), grid_lr_weight=500.0, enable_amp=True,
                             scheduler=True, prune_every=100, rgb_loss_type='huber', max_epochs=10 ** 6, target_sample_size=2 ** 18)
tr = MultiviewTrainer(cfg, pipe, ds, device=dev)
tr.is_optimization_running = True
for _ in range(30):
    tr.iterate()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    tr.iterate()
torch.cuda.synchronize()
print(f"iterate: {(time.perf_counter() - t0) * 10:.3f} ms/step, rays {ds.transform.num_samples}, samples {pipe.tracer.get_prev_num_samples()}")
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    tr.iterate()
torch.cuda.synchronize(); pr.disable()
for key in ("tottime", "cumulative"):
    buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(40); print(buf.getvalue()[:7000])
if os.environ.get("NO_TORCH_PROFILER"):
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        tr.iterate()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=70))
