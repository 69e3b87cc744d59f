"""End-to-end sanity run at bench scale: train the nerf_hash.yaml pipeline on SynLego for a few hundred steps - starting
from a DENSE level-7 octree, pruning every 100 steps like MultiviewTrainer.pre_step, adaptive ray count - and print the
held-out PSNR as it goes.  Not a benchmark (the eval renders are inside the loop); it shows that the fast path trains."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch
import bench, synlego
from wisp.accelstructs import OctreeAS
from wisp.core import Rays
from wisp.trainers import MultiviewTrainStep
from wisp.trainers.validation import evaluate_psnr

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dense = OctreeAS.make_dense(level=7).points
dense = dense[-(128 ** 3):].to(dev)                       # the level-7 cells of the dense tree
pipe = bench.build_pipeline(dev, 64, 2048, dense)
tr = MultiviewTrainStep(pipe, prune_every=100, target_sample_size=2 ** 21, max_rays=2 ** 18, enable_amp=True,
                        scheduler_milestones=[], lr=1e-3, grid_lr_weight=100.0)
o, d, rgb = synlego.ray_bank(2 ** 20, seed=1, device=dev)
eo, ed, ergb = synlego.ray_bank(2 ** 15, seed=999, device=dev)
views = [(Rays(eo, ed, dist_min=1.0, dist_max=5.0), ergb)]
R = 4096
t0 = time.perf_counter()
for it in range(1, steps + 1):
    idx = torch.randint(0, o.shape[0], (R,), device=dev)
    loss, ns = tr.step(Rays(o.index_select(0, idx), d.index_select(0, idx), dist_min=1.0, dist_max=5.0), rgb.index_select(0, idx))
    R = max(1024, tr.num_rays)
    if it % 100 == 0 or it == 1 or (os.environ.get("VERBOSE") and it % 10 == 0):
        tr.wait_for_parameters()
        p, line = (0.0, 'noeval') if os.environ.get('NOEVAL') else evaluate_psnr(pipe, views, epoch=it, max_epochs=steps, amp=True)
        print(f"step {it:4d}  rays {R:6d}  samples {ns:8d}  loss {float(loss):.5f}  cells {pipe.nef.grid.blas.points.shape[0]}  {line}  "
              f"[{time.perf_counter() - t0:.1f}s]", flush=True)
