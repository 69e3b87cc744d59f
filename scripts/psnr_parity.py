"""PSNR parity at the flagship shape: the nerf_hash.yaml model (level-7 octree DENSE at the start,
HashGrid L=16 F=2 T=2^19, hidden 64, 'ray' march, huber, AdamW with the grid lr x 100) trained for >= 1000 steps on
SynLego with a reduced ray count, pruning every 100 steps, once through the HIP path and once through the CPU oracle -
same initial weights, same ray batches, same raymarch jitter, same prune draws - logging held-out PSNR as it goes.

    python scripts/psnr_parity.py --backend hip    --out profiles/r02_psnr_parity_hip.log       (GPU box)
    python scripts/psnr_parity.py --backend oracle --out profiles/r02_psnr_parity_oracle.log    (any CPU; ~20-40 min)
    python scripts/psnr_parity.py --compare profiles/r02_psnr_parity_hip.log profiles/r02_psnr_parity_oracle.log

The two runs need not share a machine: every random draw comes from seeded numpy / torch-CPU generators.
The oracle is test infrastructure; this script is a checker, not a product path."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]

import numpy as np
import torch

NGP = dict(feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=1e-9, codebook_bitwidth=19,
           min_grid_res=16, max_grid_res=512)
LEVEL, DECAY, MIN_DENSITY = 7, 0.95, 2.956033378250884


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle-half", action="store_true",
                    help="oracle backend: the enable_amp regime - round where fp16 autocast rounds (tables and lookups half, nn.Linear on half "
                         "operands), scaled loss, fp32 unscale, skipped steps: what the dropin backend is to be compared with")
    ap.add_argument("--backend", choices=["hip", "dropin", "oracle"],
                    help="hip = MultiviewTrainStep (fused step); dropin = wisp.trainers.MultiviewTrainer, the reference trainer's own step "
                         "(fp16 autocast + GradScaler + torch.optim.AdamW over the modular pipeline); oracle = CPU restatement")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--num-steps", type=int, default=2048)
    ap.add_argument("--eval-rays", type=int, default=4096)
    ap.add_argument("--perturb", type=float, default=0.0,
                    help="relative N(0, perturb) noise on the initial decoder weights: a second run of the SAME backend with "
                         "e.g. 1e-6 measures how far two trajectories of this chaotic optimisation drift apart on their own")
    ap.add_argument("--amp", action="store_true", help="hip backend: bf16 tables + bf16 decoder (the bench's default precision) instead of fp32")
    ap.add_argument("--out", default=None)
    ap.add_argument("--compare", nargs=2, default=None)
    return ap.parse_args()


def compare(a, b):
    def read(p):
        rows = {}
        for line in open(p):
            if line.startswith("step "):
                f = line.split()
                rows[int(f[1])] = float(f[f.index("psnr") + 1])
        return rows
    ra, rb = read(a), read(b)
    worst = 0.0
    for it in sorted(set(ra) & set(rb)):
        print(f"step {it:5d}  {ra[it]:7.3f} dB  {rb[it]:7.3f} dB  diff {ra[it] - rb[it]:+.3f}")
        worst = max(worst, abs(ra[it] - rb[it]))
    last = max(set(ra) & set(rb))
    print(f"final |diff| = {abs(ra[last] - rb[last]):.3f} dB, worst over the run = {worst:.3f} dB")


def main():
    args = parse()
    if args.compare:
        return compare(*args.compare)
    import synlego
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    out = open(args.out, "w") if args.out else sys.stdout

    def log(msg):
        print(msg, file=out, flush=True)
        if out is not sys.stdout:
            print(msg, flush=True)

    # identical initial weights: the module is constructed on the host in both runs (parameters only, no kernels)
    torch.manual_seed(0)
    blas0 = OctreeAS.make_dense(LEVEL)
    grid = HashGrid.from_geometric(blas0, **NGP)
    nef = NeuralRadianceField(grid, pos_embedder='none', view_embedder='positional', view_multires=4, activation_type='relu',
                              layer_type='linear', hidden_dim=64, num_layers=1, bias=True, prune_density_decay=DECAY,
                              prune_min_density=MIN_DENSITY)
    if args.perturb > 0:
        g = torch.Generator().manual_seed(77)
        with torch.no_grad():
            for n, p in nef.named_parameters():
                if 'decoder' in n:
                    p.mul_(1.0 + args.perturb * torch.randn(p.shape, generator=g))
    o, d, gt = synlego.ray_bank(1 << 17, seed=11, device='cpu')
    eo, ed, egt = synlego.ray_bank(args.eval_rays, seed=12, device='cpu')
    rng = np.random.default_rng(2024)                       # batches + jitter
    prune_gen = torch.Generator().manual_seed(0)            # the draws MultiviewTrainStep.prune makes (seed 0)
    eval_jit = np.random.default_rng(5).uniform(size=(args.eval_rays, args.num_steps)).astype(np.float32)
    log(f"# backend={args.backend} perturb={args.perturb} eval_rays={args.eval_rays} steps={args.steps} rays/step={args.rays} candidates/ray={args.num_steps} "
        f"grid L=16 F=2 T=2^19 std={NGP['feature_std']} level-{LEVEL} dense start, prune every 100, AdamW lr 1e-3 grid x100")
    t0 = time.time()

    if args.backend == "hip":
        from wisp.core import Rays
        from wisp.models import Pipeline
        from wisp.tracers import PackedRFTracer
        from wisp.trainers import MultiviewTrainStep
        dev = torch.device("cuda", 0)
        nef = nef.to(dev)
        pipe = Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=args.num_steps, bg_color=(0.0, 0.0, 0.0)))
        tr = MultiviewTrainStep(pipe, prune_every=100, lr=1e-3, grid_lr_weight=100.0, seed=0, prune_rng_device='cpu', enable_amp=args.amp)

        def evaluate():
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=args.amp):
                rb = pipe(rays=Rays(eo.to(dev), ed.to(dev), dist_min=1.0, dist_max=5.0), channels=["rgb"],
                          jitter=torch.from_numpy(eval_jit).to(dev))
            mse = float(((rb.rgb.float().cpu() - egt) ** 2).mean())
            return 10 * np.log10(1.0 / mse), int(pipe.nef.grid.blas.pyramid[0, LEVEL])

        def step(idx, jit):
            loss, ns = tr.step(Rays(o[idx].to(dev), d[idx].to(dev), dist_min=1.0, dist_max=5.0), gt[idx].to(dev),
                               jitter=torch.from_numpy(jit).to(dev))
            return float(loss), ns
    elif args.backend == "dropin":
        # the unchanged-trainer regime on identical batches / jitter / prune draws: pre_step (prune timing) and step() of the class,
        # called the way BaseTrainer.iterate calls them (pre_step outside, step inside `torch.autocast('cuda')` = fp16)
        from wisp.core import Rays
        from wisp.datasets import MultiviewTensorDataset, SampleRays
        from wisp.models import Pipeline
        from wisp.tracers import PackedRFTracer
        from wisp.trainers import MultiviewTrainer, ConfigMultiviewTrainer, ConfigAdamW
        dev = torch.device("cuda", 0)
        nef = nef.to(dev)
        pipe = Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=args.num_steps, bg_color=(0.0, 0.0, 0.0)))
        ds = MultiviewTensorDataset(o[None, :args.rays].to(dev), d[None, :args.rays].to(dev), gt[None, :args.rays].to(dev), 1.0, 5.0,
                                    transform=SampleRays(args.rays))
        cfg = ConfigMultiviewTrainer(optimizer=ConfigAdamW(lr=1e-3, eps=1e-16, weight_decay=1e-6), grid_lr_weight=100.0, enable_amp=True,
                                     scheduler=False, prune_every=100, rgb_loss_type='huber', rgb_loss_denom='rays', max_epochs=10 ** 6)
        tr = MultiviewTrainer(cfg, pipe, ds, device=dev)
        tr.iterations_per_epoch = 10 ** 9

        def seeded_prune():                                   # the draws MultiviewTrainStep.prune makes with prune_rng_device='cpu'
            cells = nef.grid.dense_points.shape[0]
            unit = torch.rand(cells, 3, generator=prune_gen)
            views = torch.nn.functional.normalize(torch.randn(cells, 3, generator=prune_gen), dim=1)
            type(nef).prune(nef, unit_samples=unit, view_dirs=views)
        nef.prune = seeded_prune
        with torch.autocast('cuda', enabled=True):            # the trainer's first call only sizes the batch
            tr.step({"rays": Rays(o[None, :args.rays].to(dev), d[None, :args.rays].to(dev), dist_min=1.0, dist_max=5.0),
                     "rgb": gt[None, :args.rays].to(dev)})
        count = {"it": 0}

        def evaluate():
            pipe.tracer.jitter = torch.from_numpy(eval_jit).to(dev)
            with torch.no_grad():
                rb = pipe(rays=Rays(eo.to(dev), ed.to(dev), dist_min=1.0, dist_max=5.0), channels=["rgb"])
            mse = float(((rb.rgb.float().cpu() - egt) ** 2).mean())
            return 10 * np.log10(1.0 / mse), int(pipe.nef.grid.blas.pyramid[0, LEVEL])

        def step(idx, jit):
            tr.iteration = count["it"]                        # 0-based like the other two backends: prunes before steps 101, 201, ...
            count["it"] += 1
            tr.pre_step()
            pipe.tracer.jitter = torch.from_numpy(jit).to(dev)
            before = tr.tracker.metrics.rgb_loss
            with torch.autocast('cuda', enabled=True):
                tr.step({"rays": Rays(o[idx][None].to(dev), d[idx][None].to(dev), dist_min=1.0, dist_max=5.0), "rgb": gt[idx][None].to(dev)})
            return tr.tracker.metrics.rgb_loss - before, pipe.tracer.get_prev_num_samples()
    else:
        from oracle import nerf as onerf, spc as ospc
        res = [int(r) for r in grid.resolutions]
        onef = onerf.OracleNeRF(res, 2, 19, 'cat', NGP['feature_std'], 64, 1, True, 4)
        onef.load_state_dict({k: v.detach() for k, v in nef.state_dict().items() if k in onef.state_dict()}, strict=False)
        state = {"blas": onerf.OracleBLAS.make_dense(LEVEL), "occ": torch.zeros(128 ** 3), "it": 0}
        dense_points = state["blas"].level_points().copy()
        opt = onerf.make_optimizer(onef, lr=1e-3, grid_lr_weight=100.0)
        scaler = onerf.make_scaler() if args.oracle_half else None

        def evaluate():
            with torch.no_grad():
                r = onerf.trace(onef, state["blas"], eo, ed, 1.0, 5.0, args.num_steps, eval_jit, (0.0, 0.0, 0.0), 'ray', with_depth=False)
            return onerf.psnr(r["rgb"], egt), int(state["blas"].pyramid[0, LEVEL])

        def step(idx, jit):
            # MultiviewTrainStep.pre_step: prune before the step when total_iterations > 1 and divisible by 100
            it = state["it"]
            if it > 1 and it % 100 == 0:
                cells = dense_points.shape[0]
                unit = torch.rand(cells, 3, generator=prune_gen)
                views = torch.nn.functional.normalize(torch.randn(cells, 3, generator=prune_gen), dim=1)
                nb, state["occ"] = onerf.prune(onef, state["blas"], state["occ"], dense_points, DECAY, MIN_DENSITY, unit, views)
                if nb is not None:
                    state["blas"] = nb
            state["it"] += 1
            return onerf.train_step(onef, state["blas"], opt, o[idx], d[idx], gt[idx], 1.0, 5.0, args.num_steps, jit, scaler=scaler)

    for it in range(1, args.steps + 1):
        idx = torch.from_numpy(rng.integers(0, o.shape[0], args.rays))
        jit = rng.uniform(size=(args.rays, args.num_steps)).astype(np.float32)
        loss, ns = step(idx, jit)
        if it == 1 or it % 100 == 0:
            p, cells = evaluate()
            log(f"step {it} loss {loss:.6f} samples {ns} cells {cells} psnr {p:.3f} [{time.time() - t0:.0f}s]")


if __name__ == "__main__":
    main()
