"""Host-time split of one iteration of the unchanged-trainer regime (wisp.trainers.MultiviewTrainer.iterate): wall time per phase,
measured with perf_counter around the phases of step() (no extra synchronisation: what the phases themselves wait for is in them)."""
import os, sys, time, collections
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
                             scheduler=True, prune_every=-1, rgb_loss_type='huber', max_epochs=10 ** 6, target_sample_size=2 ** 18)
tr = MultiviewTrainer(cfg, pipe, ds, device=dev)
tr.is_optimization_running = True
for _ in range(60):
    tr.iterate()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        tr.iterate()
    torch.cuda.synchronize()
    print(f"iterate: {(time.perf_counter() - t0) * 5:.3f} ms/step, rays {ds.transform.num_samples}, samples {pipe.tracer.get_prev_num_samples()}")

# phase split: wrap the callables step() goes through
acc = collections.defaultdict(float)
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
    setattr(obj, name, w)
wrap(tr, "next_batch", "next_batch (loader + SampleRays)")
wrap(tr, "step", "step() total")
wrap(tr.pipeline, "forward", "pipeline forward") if False else None
orig_call = type(tr.pipeline).__call__
wrap(tr.optimizer, "zero_grad", "optimizer.zero_grad")
wrap(tr.scaler, "step", "scaler.step (unscale + inf check + AdamW)")
wrap(tr.scaler, "update", "scaler.update")
wrap(tr.scaler, "scale", "scaler.scale")
wrap(tr.scheduler, "step", "scheduler.step")
wrap(tr, "calc_adaptive_rays", "calc_adaptive_rays")
wrap(pipe.tracer, "trace", "tracer.trace (forward)")
wrap(pipe.nef.grid, "raymarch", "  grid.raymarch (incl. count read-back)")
wrap(pipe.nef, "rgba", "  nef.rgba (interpolate + decoder)")
import torch.autograd
orig_bwd = torch.Tensor.backward
def bwd(self, *a, **k):
    t = time.perf_counter()
    try:
        return orig_bwd(self, *a, **k)
    finally:
        acc["loss.backward"] += time.perf_counter() - t
torch.Tensor.backward = bwd
orig_item = torch.Tensor.item
def item(self):
    t = time.perf_counter()
    try:
        return orig_item(self)
    finally:
        acc[".item() read-backs"] += time.perf_counter() - t
torch.Tensor.item = item
N = 200
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    tr.iterate()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N * 1e3
print(f"instrumented iterate: {tot:.3f} ms/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {v / N * 1e3:7.3f} ms  {k}")
