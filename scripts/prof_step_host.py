"""cProfile of MultiviewTrainStep.step (the fused, direct-issue step) at the reference trainer's 2^18 samples per step: the GPU needs
0.35 ms for such a step and the kernel trace shows it idle ~10 % of the window, i.e. the HOST is close to critical there."""
import cProfile, os, pstats, sys, io, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaolin-wisp_amd"))
import torch
import bench, synlego
from wisp.core import Rays
from wisp.trainers import MultiviewTrainStep

dev = torch.device("cuda:0")
target = int(os.environ.get("TARGET", 2 ** 18))
cells = synlego.occupied_cells(7, device=dev)
pipe = bench.build_pipeline(dev, 64, 2048, cells)
tr = MultiviewTrainStep(pipe, lr=1e-3, eps=1e-16, weight_decay=1e-6, grid_lr_weight=500.0, rgb_loss_type='huber', prune_every=-1,
                        target_sample_size=target, max_rays=2 ** 18, enable_amp=True)
bo, bd, brgb = synlego.ray_bank(2 ** 20, seed=1, device=dev)
gen = torch.Generator(device=dev).manual_seed(3)
import wisp._C as C

def batch(n):
    idx = torch.randint(0, bo.shape[0], (n,), device=dev, generator=gen)
    o, d, rgb = C.gather_rows(idx, [bo, bd, brgb])
    return Rays(o, d, dist_min=synlego.NEAR, dist_max=synlego.FAR), rgb

R = 4096
rays, gts = batch(R)
for _ in range(40):
    nr, ng = batch(R)
    tr.step(rays, gts, prefetch=nr)
    R = tr.num_rays
    rays, gts = nr, ng
R = tr.num_rays

def loop(n):
    global rays, gts
    for _ in range(n):
        nr, ng = batch(R)
        tr.step(rays, gts, prefetch=nr)
        rays, gts = nr, ng

loop(50)
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter(); loop(N); t_issue = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"target {target}: rays/step {R}; host issue time {1e3 * t_issue / N:.3f} ms/step, wall incl. drain {1e3 * t_all / N:.3f} ms/step")
pr = cProfile.Profile(); pr.enable(); loop(N); torch.cuda.synchronize(); pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(34)
    print(f"==== by {key} ({N} steps)")
    print("\n".join(l[:160] for l in s.getvalue().splitlines()[4:46]))
