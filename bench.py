#!/usr/bin/env python
"""bench.py - training rays/sec of HashGrid NeRF (app/nerf nerf_hash.yaml shape) on the synthetic Lego stand-in.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (N > 1 without a launcher: starts the N ranks itself, the same way)

One "step" = one pass of the hot path over one batch of rays: on-device ray sampling from the bank, raymarch
('ray', 2048 candidates per ray against the level-7 occupancy octree), hash-grid interpolation (L=16, F=2, T=2^19),
density + colour MLPs, packed compositing, huber loss, backward, (RCCL all-reduce when N>1), fused AdamW, and - every
100th step, wherever that falls - the occupancy prune (nerf.py:175-212).
Prints ONE JSON line (rank 0).  Inputs are synthetic (synlego.py) and resident in HBM before the timed region.

Sequence: dense level-7 octree (nerf_hash.yaml:16-17) -> `--pretrain` untimed optimisation steps with the trainer's own
adaptive ray count and pruning -> W warm-up + exactly K timed steps at `--target-samples` packed samples per step - the
reference trainer's own batch, 2^18 (multiview_trainer.py:58): the headline `value` -> W + K steps at `--large-target-samples`
(2^21, 8 x the reference's) with the learning rates scaled by sqrt(batch ratio) (`large_batch_regime`: more rays per second,
less quality per ray - `quality` holds the held-out PSNR of both regimes against rays consumed and against wall-clock,
profiles/r06_time_to_psnr_*.txt) -> the unchanged trainer (`dropin_regime`) -> one prune timed on its own (`prune`) -> PSNR on
held-out rays -> live rocprofv3 --pmc passes for `roofline.traffic` -> the N > 1 launch structure on one GPU
(`dp_path_regime`) -> CPU oracle baseline.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "kaolin-wisp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

NGP = dict(feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=1e-9, codebook_bitwidth=19,
           min_grid_res=16, max_grid_res=512)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pretrain", type=int, default=300,
                    help="untimed optimisation steps from the DENSE level-7 octree before warm-up (prune every 100, adaptive "
                         "ray count): the timed steps then run on an occupancy the model has learned, and the PSNR means something")
    ap.add_argument("--num-steps", type=int, default=2048, help="raymarch candidates per ray (nerf_hash.yaml:49)")
    ap.add_argument("--target-samples", type=int, default=2 ** 18,
                    help="packed samples per step per GPU of the headline regime: the reference trainer's target_sample_size "
                         "(multiview_trainer.py:58)")
    ap.add_argument("--large-target-samples", type=int, default=2 ** 21,
                    help="packed samples per step per GPU of large_batch_regime (the headline until round 5; 0 skips it)")
    ap.add_argument("--large-lr-scale", type=float, default=None,
                    help="learning-rate multiplier of large_batch_regime; default sqrt(large / target) - the rule "
                         "profiles/r06_time_to_psnr_*.txt measured (x 2-3 at 8 x the batch)")
    ap.add_argument("--quality-budget", type=float, default=4.096e7,
                    help="rays each regime trains a fresh model on for `quality` (held-out PSNR at equal rays and equal seconds); "
                         "default = the reference's 100 epochs x 100 views x 4096 rays; 0 skips it")
    ap.add_argument("--bank-rays", type=int, default=2 ** 21)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--occupancy", choices=["dense", "analytic"], default="dense",
                    help="'dense' = nerf_hash.yaml:16-17 (make_dense level 7, pruned while training); 'analytic' = start from "
                         "the scene's true occupied cells (kernel profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--eval-rays", type=int, default=2 ** 16)
    ap.add_argument("--config", choices=["nerf_hash", "v8", "vqad", "nglod"], default="nerf_hash",
                    help="nerf_hash = BASELINE.json's metric configuration (C2; the driver's line); v8 / vqad / nglod = the other "
                         "configs on their synthetic stand-ins (bench_configs.py), one GPU, secondary lines")
    ap.add_argument("--sdf-batch", type=int, default=512, help="nglod: coordinates per step (nglod_octree.yaml:78)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short secondary-configuration runs (`configs` in the line)")
    ap.add_argument("--dropin-steps", type=int, default=100,
                    help="timed iterations of the reference trainer's own step (fp16 autocast + GradScaler + torch.optim), reported "
                         "as dropin_regime; 0 skips it")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --target-samples per GPU (per-GPU work fixed as N grows); strong: --target-samples is the GLOBAL "
                         "batch, every rank takes 1/N of it (the reference trainer's batch stays the reference's as GPUs are added)")
    ap.add_argument("--dp-steps", type=int, default=30,
                    help="one GPU only: timed steps of the N > 1 launch structure (gradient collective on a side stream + separate "
                         "optimizer) over a one-rank RCCL group, reported as dp_path_regime; 0 skips it")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def build_pipeline(dev, hidden, num_steps, blas_cells):
    from wisp.accelstructs import OctreeAS
    from wisp.models import Pipeline
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    torch.manual_seed(0)                                   # identical replicas on every rank
    blas = OctreeAS.from_quantized_points(blas_cells.to(dev), 7)
    grid = HashGrid.from_geometric(blas, **NGP)
    nef = NeuralRadianceField(grid, pos_embedder='none', view_embedder='positional', view_multires=4,
                              activation_type='relu', layer_type='linear', hidden_dim=hidden, num_layers=1, bias=True,
                              prune_density_decay=0.95, prune_min_density=2.956033378250884).to(dev)
    tracer = PackedRFTracer(raymarch_type='ray', num_steps=num_steps, step_size=1.0, bg_color=(0.0, 0.0, 0.0))
    return Pipeline(nef, tracer)


def cpu_baseline(blas_cells, hidden, num_steps, budget_s=25.0, threads=16):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores: same model shape, same occupancy.
    BASELINE.md section 3 names R = 4096 rays per step; one such oracle step is 8.4 M candidates and takes 10 - 100 s of host
    time depending on the box, so the bounded sample runs the SAME step at R = 256 (or 128 / 64 on a slow host) - rays/s of the
    oracle is flat in R (every stage is per-ray or per-sample work) - for at least 10 steps within the budget.  The intra-op
    thread count is PINNED (16): with torch's default (= all hardware threads, 128 on the driver's box) the small per-level ops
    spend their time in thread wake-ups and the figure swung 7x between boxes (43.9 vs 312 rays/s in round 3)."""
    from oracle import nerf as onerf, spc as ospc
    import synlego
    torch.manual_seed(0)
    had = torch.get_num_threads()
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or threads)))
    try:
        res = [int(np.floor(16 * (np.exp((np.log(512) - np.log(16)) / 15) ** l))) for l in range(16)]
        nef = onerf.OracleNeRF(res, 2, 19, 'cat', 1e-9, hidden, 1, True, 4)
        blas = onerf.OracleBLAS(ospc.points_to_octree(blas_cells.cpu().numpy(), 7))
        opt = onerf.make_optimizer(nef)
        rng = np.random.default_rng(0)

        def step(o, d, gts):
            return onerf.train_step(nef, blas, opt, o, d, gts, 1.0, 5.0, num_steps,
                                    rng.uniform(size=(o.shape[0], num_steps)).astype(np.float32))[1]

        R = 256
        o, d, _ = synlego.ray_bank(R, seed=123, device='cpu', with_gt=False)
        gts = torch.rand(R, 3)
        step(o, d, gts)                                       # warm-up (allocations, thread pool)
        t0 = time.time()
        step(o, d, gts)
        one = time.time() - t0
        while R > 64 and one * 10 > budget_s:                 # at least 10 steps must fit the budget
            R //= 2
            one /= 2
        o, d, gts = o[:R], d[:R], gts[:R]
        t0, n, samples = time.time(), 0, 0
        while n < 10 or (time.time() - t0) < budget_s:
            samples += step(o, d, gts)
            n += 1
            if n >= 60:
                break
        dt = time.time() - t0
        return dict(value=R * n / dt, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                    host_cpus=os.cpu_count(), torch_num_threads=torch.get_num_threads(),
                    sample=f"{n} oracle train steps of {R} rays x {num_steps} candidates ({samples // max(n, 1)} packed samples/step), "
                           f"fp32, torch-CPU + numpy, {torch.get_num_threads()} intra-op threads pinned, {dt:.1f}s; BASELINE.md's R = 4096 "
                           f"step is the same work x {4096 // R} (not run: it alone would exceed the bench's time budget)")
    finally:
        torch.set_num_threads(had)


def collective_selftest(dev, rank):
    """Every collective the training step will issue, once, on small tensors, checked against host arithmetic - BEFORE anything
    is timed: the first multi-GPU run of this code must not be the first time RCCL sees these calls.  All ranks agree on the
    verdict (an all-reduce of the flags).  -> dict(rccl_ranks, allreduce_ok, sharded_path_ok)."""
    world = dist.get_world_size()
    tri = world * (world + 1) / 2.0
    ok = {}
    x = torch.full((1024,), float(rank + 1), device=dev)
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    ok["allreduce_ok"] = bool((x == tri).all())
    try:
        src = torch.arange(world * 256, device=dev, dtype=torch.float32) * float(rank + 1)
        got = torch.empty(256, device=dev)
        dist.reduce_scatter_tensor(got, src, op=dist.ReduceOp.SUM)
        want = torch.arange(rank * 256, (rank + 1) * 256, device=dev, dtype=torch.float32) * tri
        rs = bool(torch.equal(got, want))
        own = torch.full((256,), float(rank), device=dev, dtype=torch.bfloat16)
        full = torch.empty(world * 256, device=dev, dtype=torch.bfloat16)
        dist.all_gather_into_tensor(full[:world * 256], own)
        ag = bool(torch.equal(full.float().view(world, 256), torch.arange(world, device=dev, dtype=torch.float32)[:, None].expand(world, 256)))
        ok["sharded_path_ok"] = rs and ag
    except Exception as e:                                   # a hard RCCL failure would abort the process; this catches the soft ones
        ok["sharded_path_ok"] = False
        ok["sharded_path_error"] = f"{type(e).__name__}: {e}"
    flags = torch.tensor([float(ok["allreduce_ok"]), float(ok["sharded_path_ok"])], device=dev)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    ok["allreduce_ok"], ok["sharded_path_ok"] = bool(flags[0] > 0), bool(flags[1] > 0)
    ok["rccl_ranks"] = world
    assert ok["allreduce_ok"], "RCCL all-reduce returned wrong sums"
    return ok


def hidden128_line(args, dev, pipe, batch, target_samples, steps=30):
    """nerf_hash.yaml with hidden_dim 128 - the reference's best published row (docs/pages/app_nerf.md:185-192) - on the
    occupancy the main run has learned: a fresh model of that width over a copy of the current octree, a few steps at the
    given batch size (direct-issue step under amp, like the headline)."""
    from wisp.accelstructs import OctreeAS
    from wisp.models import Pipeline
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep
    import wisp._C as C
    try:
        blas = OctreeAS(pipe.nef.grid.blas.octree.clone())
        torch.manual_seed(0)
        grid = HashGrid.from_geometric(blas, **NGP)
        nef = NeuralRadianceField(grid, pos_embedder='none', view_embedder='positional', view_multires=4, activation_type='relu',
                                  layer_type='linear', hidden_dim=128, num_layers=1, bias=True, prune_density_decay=None,
                                  prune_min_density=None).to(dev)
        wide = Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=args.num_steps, step_size=1.0, bg_color=(0.0, 0.0, 0.0)))
        tr = MultiviewTrainStep(wide, lr=1e-3, eps=1e-16, weight_decay=1e-6, grid_lr_weight=500.0, rgb_loss_type='huber',
                                prune_every=-1, target_sample_size=target_samples, max_rays=2 ** 18, enable_amp=args.precision == "bf16")
        probe, _ = batch(4096)
        rm = grid.raymarch(probe, level=grid.active_lods[-1], num_samples=args.num_steps, raymarch_type='ray')
        wide.tracer.prev_num_samples = rm.samples.shape[0]
        R = max(256, tr.calc_adaptive_rays(4096))
        state = {"rays": None, "gts": None}
        state["rays"], state["gts"] = batch(R)

        def loop(n):                                  # one-batch look-ahead, like the headline's timed loop
            total = 0
            for _ in range(n):
                nrays, ngts = batch(R)
                _, ns = tr.step(state["rays"], state["gts"], prefetch=nrays)
                total += ns
                state["rays"], state["gts"] = nrays, ngts
            return total

        loop(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        samples = loop(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # per-launch events in a second, short pass: a pair of events around every launch is not free (see bench_configs._nerf_run)
        C.TIMING_ALL = {}
        loop(8)
        torch.cuda.synchronize()
        sink, C.TIMING_ALL = C.TIMING_ALL, None
        k = {n.replace("wisp_", ""): float(np.mean([a.elapsed_time(b) for a, b in ev])) for n, ev in sink.items()}
        S = samples / steps
        flop = 2 * (32 * 128 + 16 * 128 + 42 * 128 + 128 * 128 + 3 * 128)              # 56 576 per sample forward
        roof = {}
        for name, mult in (("nerf_mlp_fwd", 1), ("nerf_mlp_bwd", 3)):
            if name in k:
                tf = mult * flop * S / (k[name] * 1e-3) / 1e12
                roof[name] = {"avg_ms": k[name], "bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0}
        return {"metric": "training rays/sec, HashGrid NeRF hidden 128", "value": R * steps / dt, "unit": "rays/s",
                "ms_per_step": 1e3 * dt / steps, "steps": steps, "rays_per_step": R, "samples_per_step": S, "roofline": roof,
                "top_launches": dict(sorted(k.items(), key=lambda kv: -kv[1])[:5])}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def quality_table(args, dev, train_bank, amp, large_lr):
    """`quality`: held-out PSNR of a FRESH nerf_hash.yaml model trained under the headline regime and under large_batch_regime on
    the same ray budget (default: the reference's 100 epochs = 4.1e7 rays) - against rays consumed and against training seconds
    (bench_quality.time_to_psnr; evaluation excluded from the seconds).  BASELINE.json's metric is rays/sec + PSNR: this is the
    PSNR half, per regime."""
    import bench_quality as bq
    import synlego
    try:
        eval_bank = synlego.ray_bank(2 ** 15, seed=7, device=dev)
        budget = int(args.quality_budget)
        marks = tuple(m for m in (1e7, 2e7, 4e7) if m <= budget) or (float(budget) * (1 - 1e-9),)
        out = {"ray_budget": budget, "held_out_rays": int(eval_bank[0].shape[0]), "train_bank_rays": int(train_bank[0].shape[0]),
               "bank_note": "the bench's resident ray bank (--bank-rays, 2^21 by default) is drawn ~20 times over by a 4.1e7-ray budget; "
                            "profiles/r06_time_to_psnr_*.txt use a 2^23-ray bank and end ~4 dB higher in every regime - the ORDER of the "
                            "regimes is the same",
               "note": "fresh model per regime, dense level-7 start, prune every 100 steps, MultiStepLR x0.333 at 50/75/90 % of the ray "
                       "budget, same ray stream and seed; curve rows = (rays consumed, optimizer steps, training seconds, dB)"}
        for key, target, lr in (("headline", args.target_samples, 1.0), ("large_batch_regime", args.large_target_samples, large_lr)):
            if target <= 0:
                continue
            r = bq.time_to_psnr(dev, dict(target=target, accum=1), train_bank, eval_bank, ray_budget=budget, checkpoints=marks,
                                hidden=args.hidden, num_steps=args.num_steps, amp=amp, lr_scale=lr,
                                log_every_rays=budget / 8)
            out[key] = {"target_samples_per_step": target, "lr_scale": lr, "psnr_at_rays": r["psnr_at_rays"],
                        "optimizer_steps": r["optimizer_steps"], "train_seconds": r["train_seconds"],
                        "final_psnr_db": r["curve"][-1][3] if r["curve"] else None, "curve": [list(c) for c in r["curve"]]}
        return out
    except Exception as e:                                  # a quality side-run must not take the headline down
        return {"error": f"{type(e).__name__}: {e}"}


def dp_path_regime(args, dev, pipe, batch, R, headline_ms):
    """One GPU: the launch structure a rank runs when N > 1 - gradient collective + optimizer on a side stream, the table's AdamW
    NOT folded into the backward, the next step's raymarch overlapping the exchange - over a ONE-rank RCCL group
    (WISP_FORCE_ALLREDUCE=1), on a copy of the trained model at the headline batch: both exchange paths (all-reduce + replicated
    optimizer; reduce-scatter + optimizer on the own slice + all-gather of the bf16 shadow).  With one rank RCCL's collectives
    are device-local copies, so what this measures is everything of the N > 1 step EXCEPT the wire: the only unknown left for an
    8-GPU node is xGMI itself.  `projected` adds the wire from the link rate alone and is labelled as a projection."""
    import copy
    import socket
    from wisp.trainers import MultiviewTrainStep
    out = {}
    try:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    except Exception as e:
        return {"error": f"one-rank process group: {type(e).__name__}: {e}"}
    had = os.environ.get("WISP_FORCE_ALLREDUCE")
    os.environ["WISP_FORCE_ALLREDUCE"] = "1"
    try:
        for name, sharded in (("allreduce", False), ("sharded", True)):
            twin = copy.deepcopy(pipe)
            tr = MultiviewTrainStep(twin, lr=1e-3, eps=1e-16, weight_decay=1e-6, grid_lr_weight=500.0, rgb_loss_type='huber',
                                    prune_every=-1, target_sample_size=args.target_samples, max_rays=2 ** 18,
                                    enable_amp=args.precision == "bf16", sharded_optimizer=sharded)
            assert tr.force_allreduce and tr.sharded_optimizer == sharded
            rays, gts = batch(R)
            for _ in range(5):
                nrays, ngts = batch(R)
                tr.step(rays, gts, prefetch=nrays)
                rays, gts = nrays, ngts
            tr.comm_timing = []
            _sync()
            t0 = time.perf_counter()
            for _ in range(args.dp_steps):
                nrays, ngts = batch(R)
                tr.step(rays, gts, prefetch=nrays)
                rays, gts = nrays, ngts
            _sync()
            dt = time.perf_counter() - t0
            comm = tr.comm_summary() or {}
            tr.comm_timing = None
            wire = 4 * min(tr._live_grad_numel(), tr.flat.grad.numel())
            ms = 1e3 * dt / args.dp_steps
            # wire time at N = 8 from the link rate alone (MI355X_MICROARCH.md: 7 xGMI links x ~153 GB/s per GPU, fully connected):
            # a direct reduce-scatter / all-gather sends 7/8 of the buffer over 7 links in parallel = bytes / 8 / 153e9 per phase;
            # a ring is bound by ONE link: 2 x 7/8 x bytes / 153e9
            back = 2 if sharded else 4                                   # bytes per element on the way back (bf16 shadow / fp32 sum)
            direct_us = 1e6 * (wire / 8 / 153e9 + wire * back / 4 / 8 / 153e9)
            ring_us = 1e6 * (7 / 8) * (wire + wire * back / 4) / 153e9
            hidden = comm.get("hidden_ms", 0.0)
            out[name] = {"ms_per_step": ms, "value": R * args.dp_steps / dt, "unit": "rays/s", "steps": args.dp_steps,
                         "over_headline_ms": ms - headline_ms, "comm": comm, "grad_bytes_per_step": wire,
                         "projected_n8": {
                             "wire_us_direct": direct_us, "wire_us_ring": ring_us,
                             "rays_per_sec_direct": 8 * R / ((ms + max(0.0, direct_us * 1e-3 - hidden)) * 1e-3),
                             "rays_per_sec_ring": 8 * R / ((ms + max(0.0, ring_us * 1e-3 - hidden)) * 1e-3),
                             "note": "PROJECTION, not a measurement: this one-GPU step + the part of the wire time (link rate only, no "
                                     "latency, no RCCL protocol overhead) that does not fit the window the overlap hides today "
                                     "(`comm.hidden_ms`); weak scaling, 8 x this GPU's rays"}}
            del tr, twin
            torch.cuda.empty_cache()
        out["note"] = ("N > 1 launch structure on one GPU (one-rank RCCL group): the table's AdamW is a separate launch again and the "
                       "collective + optimizer run on a side stream under the next step's raymarch; `over_headline_ms` is what that "
                       "structure costs against the one-GPU step whose optimizer is folded into the backward")
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        if had is None:
            os.environ.pop("WISP_FORCE_ALLREDUCE", None)
        else:
            os.environ["WISP_FORCE_ALLREDUCE"] = had
    return out


def dropin_regime(pipe, bank_o, bank_d, bank_rgb, args, world, dev):
    """wisp.trainers.MultiviewTrainer - the mirror of the reference class, equal to its method bodies on the CPU
    (tests/test_reference_modules.py::test_dropin_trainer_class_equals_the_reference_methods) - configured like nerf_hash.yaml's
    `trainer:` block and driven through iterate() on a deep copy of the current model (learned occupancy, trained weights)."""
    import copy
    import synlego
    from wisp.datasets import MultiviewTensorDataset, SampleRays
    from wisp.trainers import MultiviewTrainer, ConfigMultiviewTrainer, ConfigAdamW
    views = 8
    per_view = bank_o.shape[0] // views
    shape = (views, per_view, 3)
    ds = MultiviewTensorDataset(bank_o[:views * per_view].view(shape), bank_d[:views * per_view].view(shape),
                                bank_rgb[:views * per_view].view(shape), synlego.NEAR, synlego.FAR, transform=SampleRays(4096))
    cfg = ConfigMultiviewTrainer(optimizer=ConfigAdamW(lr=1e-3, eps=1e-16, weight_decay=1e-6), grid_lr_weight=500.0, enable_amp=True,
                                 scheduler=True, prune_every=100, rgb_loss_type='huber', rgb_loss_denom='rays', max_epochs=10 ** 6,
                                 target_sample_size=args.target_samples)
    twin = copy.deepcopy(pipe)
    twin.tracer.prev_num_samples = None                       # the trainer's first call is its warm-up raymarch
    tr = MultiviewTrainer(cfg, twin, ds, device=dev)
    tr.is_optimization_running = True
    # warm-up past the trainer's first prune (iteration 100): its first call pays one-time allocations that are not step cost;
    # the timed iterations then contain the steady-state prunes, one per hundred
    for _ in range(102 + min(args.warmup, 10)):
        tr.iterate()
    # The trainer's first iterations import half of torch's optional machinery (torch.optim -> sympy ...): ~170 K long-lived Python
    # objects, after which the cyclic collector owes a full collection - 60-90 ms, i.e. +0.6-0.9 ms per iteration when it lands in
    # a 100-iteration window (it did in about half of the runs: the regime's "slow mode" until round 5,
    # profiles/r05_dropin_gc_pause.txt).  A training run pays it once; its set-up garbage is collected here and what is alive now
    # moves to the permanent generation (what an app does with gc.freeze() after building its trainer).  The collector stays ON:
    # every collection inside the window is counted and timed.
    import gc
    gc.collect()
    gc.freeze()
    pauses, t_gc = [], [0.0]

    def on_gc(phase, info):
        if phase == "start":
            t_gc[0] = time.perf_counter()
        else:
            pauses.append(time.perf_counter() - t_gc[0])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    gc.callbacks.append(on_gc)
    t0 = time.perf_counter()
    rays = samples = 0
    for _ in range(args.dropin_steps):
        rays += ds.transform.num_samples
        tr.iterate()
        samples += twin.tracer.get_prev_num_samples()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.callbacks.remove(on_gc)
    gc.unfreeze()
    if world > 1:
        t = torch.tensor([dt, rays, samples], device=dev, dtype=torch.float64)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        dt, rays, samples = float(t[0]), int(t[1]), int(t[2])
    return {"value": rays / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt / args.dropin_steps, "steps": args.dropin_steps,
            "rays_per_step_per_gpu": rays / args.dropin_steps / world, "samples_per_step_per_gpu": samples / args.dropin_steps / world,
            "loss_scale_at_end": float(tr.scaler.get_scale()), "dtype": "fp16 autocast + GradScaler (tables fp16, decoder bf16 MFMA)",
            "python_gc_inside_the_window": {"collections": len(pauses), "pause_ms": 1e3 * sum(pauses), "longest_ms": 1e3 * max(pauses or [0.0]),
                                            "note": "gc.collect() + gc.freeze() after the warm-up iterations; the collector runs during the window"},
            "note": "wisp.trainers.MultiviewTrainer.iterate(): the reference trainer's own step semantics (multiview_trainer.py:111-180, "
                    "base_trainer.py:205-246,316-342) - autograd over the pipeline (PackedRFTracer.trace differentiates lookup + decoder + "
                    "compositing as one node), torch.optim.AdamW, GradScaler, MultiStepLR, SampleRays, loss .item() read-backs, the prune "
                    "every 100 iterations inside the timed ones; no gradient all-reduce (the reference has none): with N > 1 this is N "
                    "independent replicas"}


# algorithmic work per packed sample (SURVEY.md 8d / DESIGN.md 4): bytes for the HBM-bound kernels, flops for the decoder
def work_table(amp, hidden=64, levels=16, live_levels=None):
    """`live_levels`: the levels the kernels really process.  PackedRFTracer queries lod_idx = num_lods - 1 and 'cat' zeroes the
    columns from lod_idx * F on (reference hash_grid.py:226-229), so with nerf_hash.yaml the FINEST level is neither gathered nor
    given a gradient: 15 of 16 levels are work done, and only those are charged (556 / 1036 B per sample instead of SURVEY's
    nominal 588 / 1100).  The output / incoming-gradient row keeps all `levels` columns (the zero columns are written / read).
    The decoder's first layer multiplies the zero columns like any other: its flops are not reduced."""
    b = 2 if amp else 4
    live = levels if live_levels is None else min(int(live_levels), levels)
    mlp_flop = 2 * (32 * hidden + 16 * hidden + 42 * hidden + hidden * hidden + 3 * hidden)     # 20 096 at hidden 64
    return {"hashgrid_fwd": ("hbm", 12 + live * 8 * 2 * b + levels * 2 * b),   # coords + 8 corners x F=2 per live level + the output row
            # 12 + L*F*b + 2*L*2^d*F*b_acc with b_acc = the table element size (1100 B for 16 levels of 16-bit tables).
            # The kernels accumulate in fp32 / 64-bit fixed point and merge runs, so what they actually move is `traffic`.
            "hashgrid_bwd": ("hbm", 12 + levels * 2 * b + 2 * live * 8 * 2 * b),
            "nerf_mlp_fwd": ("mfma", mlp_flop), "nerf_mlp_bwd": ("mfma", 3 * mlp_flop)}


PMC_KERNELS = {"hashgrid_fwd": ["hashgrid_fwd_kernel"],
               "hashgrid_bwd": ["hashgrid_bwd_emit_kernel", "hashgrid_bwd_emit_q_kernel", "hashgrid_bwd_reduce_kernel", "hashgrid_bwd_kernel"],
               "nerf_mlp_fwd": ["mlp_fwd_kernel"], "nerf_mlp_bwd": ["mlp_bwd_kernel", "nerf_mlp_reduce_kernel"]}


def live_pmc_traffic(args, kernel_groups, timeout_s=240):
    """HBM-side bytes per launch, measured NOW, for every group of kernel names in `kernel_groups` {label: [names]}: this same
    script is re-run under rocprofv3 in its --pmc-child mode (a few steps of the same step on the scene's analytic occupancy),
    one counter group per pass as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; kernel-trace
    only, nothing else).  FETCH_SIZE is doubled (gfx950 tallies 128-B requests as 64 B); both counters are KiB.  A third pass
    reads the L2's own request counters (TCC_HIT_sum / TCC_MISS_sum).
    -> ({label: {"bytes": fetch x2 + write, "fetch": ..., "write": ..., "l2_hit": ..., "l2_miss": ...}}, note)"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    names = sorted({n for g in kernel_groups.values() for n in g})
    rx = "|".join(names)
    per = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("TCC_HIT_sum", "TCC_MISS_sum")):
        out = tempfile.mkdtemp(prefix="wisp_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--kernel-include-regex", rx, "--output-format", "csv", "-d", out,
               "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--precision", args.precision,
               "--hidden", str(args.hidden), "--num-steps", str(args.num_steps), "--target-samples", str(args.target_samples)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
        except (subprocess.TimeoutExpired, OSError) as e:
            shutil.rmtree(out, ignore_errors=True)
            return None, f"rocprofv3 --pmc {counters}: {type(e).__name__}"
        got = False
        for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    c = row.get("Counter_Name")
                    if c not in counters:
                        continue
                    for kn in names:
                        if kn in row.get("Kernel_Name", ""):
                            per.setdefault(kn, {}).setdefault(c, []).append(float(row["Counter_Value"]))
                            got = True
        shutil.rmtree(out, ignore_errors=True)
        if not got and counters[0] in ("FETCH_SIZE", "WRITE_SIZE"):
            return None, f"rocprofv3 --pmc {counters}: no rows (rc {r.returncode})"
    res = {}
    for label, group in kernel_groups.items():
        # the child runs warm-up + timed steps; every dispatch of a kernel does the same work
        m = lambda kn, c: float(np.mean(per[kn][c])) if kn in per and c in per[kn] else 0.0      # noqa: E731
        fetch = sum(m(kn, "FETCH_SIZE") for kn in group) * 1024.0 * 2.0
        write = sum(m(kn, "WRITE_SIZE") for kn in group) * 1024.0
        res[label] = {"bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                      "l2_hit": sum(m(kn, "TCC_HIT_sum") for kn in group), "l2_miss": sum(m(kn, "TCC_MISS_sum") for kn in group),
                      "per_kernel": {kn: {c: float(np.mean(v)) for c, v in per[kn].items()} for kn in group if kn in per}}
    return res, ("live rocprofv3 --pmc passes of this command (--pmc-child): FETCH_SIZE x2 + WRITE_SIZE = bytes per launch on the "
                 "memory side of the L2s (Infinity-Cache hits included - the guide's HBM/rocprofv3 section), TCC_HIT/MISS = L2 requests")


# ---- the few places where main() touches the device runtime, as module-level functions: tests/test_bench_multirank.py replaces
# them (gloo, CPU tensors, a stand-in pipeline) to drive main()'s world > 1 control flow without a GPU.  Nothing else uses them.
def _device(local):
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local)
    return torch.device("cuda", local)


def _bind_near_gpu(local):
    """this rank's host threads onto the CPUs of its GPU's NUMA node (wisp._C.bind_host_near_device) -> what was bound, or None"""
    if not torch.cuda.is_available():
        return None
    import wisp._C as C
    return C.bind_host_near_device(local)


def _device_count():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run environment: start the N ranks ourselves - this very
    script under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` - instead of silently
    measuring one GPU and labelling it N.  Rank 0's JSON line passes through on stdout; the exit code is the launcher's."""
    import socket
    import subprocess
    have = _device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0])] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    rc = subprocess.run(cmd, env=env).returncode
    if rc != 0:
        raise SystemExit(rc)
    return None


def _init_dist(dev):
    dist.init_process_group(backend="nccl", device_id=dev)              # nccl == RCCL on ROCm


def _sync():
    torch.cuda.synchronize()


SENTINEL_ELEMS = 1_000_003            # x (tag + 1) float64 elements: a fill launch no other code of the run issues
REGIME_TAGS = {"headline": 0, "large_batch_regime": 1, "dropin_regime": 2}


class _regime:
    """Brackets one timed regime for the profiler: a roctx range (torch.cuda.nvtx is roctx on ROCm; `rocprofv3 --marker-trace`)
    and - with WISP_BENCH_SENTINELS=1, set by scripts/regime_stats.sh - one float64 fill launch of a size unique to the regime on
    either side, OUTSIDE the timed window, from which scripts/regime_stats.py cuts a kernel trace of the whole command into one
    kernel-stats table per regime."""

    def __init__(self, name, dev):
        self.name, self.dev = name, dev
        self.on = os.environ.get("WISP_BENCH_SENTINELS", "0") == "1" and dev.type == "cuda"

    def _mark(self):
        if self.on:
            torch.empty(SENTINEL_ELEMS * (REGIME_TAGS[self.name] + 1), dtype=torch.float64, device=self.dev).fill_(0.0)
            torch.cuda.synchronize()

    def __enter__(self):
        self._mark()
        torch.cuda.nvtx.range_push("bench:" + self.name)
        return self

    def __exit__(self, *exc):
        torch.cuda.nvtx.range_pop()
        self._mark()
        return False


def _gather_rows(idx, tensors):
    import wisp._C as C
    return C.gather_rows(idx, tensors)                                   # SampleRays: one launch for the three gathers


def _initial_cells(args, dev, true_cells):
    from wisp.accelstructs import OctreeAS
    if args.occupancy == "dense":
        return OctreeAS.make_dense(level=7).points[-(128 ** 3):].to(dev)     # nerf_hash.yaml:16-17
    return true_cells


def _probe_samples(pipe, probe, num_steps):
    """packed samples a raymarch-only pass over `probe` yields (MultiviewTrainer.step's first call, multiview_trainer.py:119-122)"""
    grid = pipe.nef.grid
    return grid.raymarch(probe, level=grid.active_lods[-1], num_samples=num_steps, raymarch_type='ray').samples.shape[0]


def _leaf_cells(pipe):
    blas = pipe.nef.grid.blas
    return int(blas.pyramid[0, blas.max_level])                          # leaf cells of the (pruned) octree


class _StdoutForTheLineOnly:
    """The contract is ONE JSON line on stdout.  Libraries write there too - RCCL prints a version banner from C stdio when a process
    group comes up, and its buffered tail is flushed at exit, i.e. AFTER a line printed from Python - so file descriptor 1 points at
    stderr for the whole run and is handed back, with the C buffers flushed, only for the line itself."""

    def __enter__(self):
        sys.stdout.flush()
        try:
            self.saved = os.dup(1)
            os.dup2(2, 1)
        except OSError:
            self.saved = None                       # (no real descriptor behind stdout: a captured stream - nothing to protect)
        return self

    def emit(self, text):
        self.restore()
        print(text, flush=True)

    def restore(self):
        if self.saved is not None:
            sys.stdout.flush()
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)      # whatever C code left in its stdio buffers goes where it was meant to go
            except Exception:
                pass
            os.dup2(self.saved, 1)
            os.close(self.saved)
            self.saved = None

    def __exit__(self, *exc):
        self.restore()
        return False


def main(argv=None):
    with _StdoutForTheLineOnly() as guard:
        return _main(argv, guard)


def _main(argv, guard):
    args = parse(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ       # started by torch.distributed.run
    if args.gpus > 1 and not launched and world == 1:
        guard.restore()                                    # (the ranks started below own stdout now: rank 0's line passes through)
        return _self_launch(args, argv)                    # N ranks were asked for and nobody started them: do it here
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, launched by torch.distributed.run"
    dev = _device(local)
    host_binding = _bind_near_gpu(local)
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _init_dist(dev)
    selftest = collective_selftest(dev, rank) if dist.is_initialized() else None
    if selftest is not None and not selftest["sharded_path_ok"]:
        os.environ["WISP_SHARDED_OPTIM"] = "0"             # reduce-scatter / all-gather misbehave here: keep the all-reduce path

    import synlego
    import wisp._C as C
    from wisp.accelstructs import OctreeAS
    from wisp.core import Rays
    from wisp.trainers import MultiviewTrainStep

    if args.config != "nerf_hash":
        import bench_configs
        assert world == 1, "the secondary configs are single-GPU lines"
        return bench_configs.main(args, dev, emit=guard.emit)
    if args.pmc_child:                                     # profiling child: fixed occupancy, a handful of steps, no extras
        args.occupancy, args.pretrain, args.warmup, args.steps = "analytic", 0, 2, 4
    global_target = args.target_samples
    if args.scaling == "strong" and world > 1:
        # the global batch is fixed: every rank takes 1/N of the packed-sample target (and so ~1/N of the rays)
        args.target_samples = max(4096, args.target_samples // world)
        args.large_target_samples = max(4096, args.large_target_samples // world) if args.large_target_samples > 0 else 0
    large_lr = args.large_lr_scale if args.large_lr_scale is not None else \
        math.sqrt(max(1.0, args.large_target_samples / max(args.target_samples, 1)))
    true_cells = synlego.occupied_cells(7, device=dev)
    cells = _initial_cells(args, dev, true_cells)
    pipe = build_pipeline(dev, args.hidden, args.num_steps, cells)
    amp = args.precision == "bf16"
    trainer = MultiviewTrainStep(pipe, lr=1e-3, eps=1e-16, weight_decay=1e-6, grid_lr_weight=500.0, rgb_loss_type='huber',
                                 prune_every=100, target_sample_size=args.target_samples, max_rays=2 ** 18, enable_amp=amp)

    # ---- ray bank resident in HBM (this rank's shard of the training rays) + ground truth
    bank_o, bank_d, bank_rgb = synlego.ray_bank(args.bank_rays, seed=1000 + rank, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    def batch(n):
        idx = torch.randint(0, bank_o.shape[0], (n,), device=dev, generator=gen)
        o, d, rgb = _gather_rows(idx, [bank_o, bank_d, bank_rgb])
        return Rays(o, d, dist_min=synlego.NEAR, dist_max=synlego.FAR), rgb

    def common_rays(n):
        """every rank uses the same ray count, so that the mean of the per-rank mean losses is the global mean"""
        if world > 1:
            t = torch.tensor([n], device=dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            n = int(t.item())
        return max(int(n), 256)

    def size_batch(target):
        """MultiviewTrainer.step's first call (multiview_trainer.py:119-122): a raymarch-only pass sizes the batch."""
        trainer.target_sample_size = target
        probe, _ = batch(4096)
        pipe.tracer.prev_num_samples = _probe_samples(pipe, probe, args.num_steps)
        return common_rays(trainer.calc_adaptive_rays(4096))

    # ---- untimed pre-training from the dense octree: adaptive ray count every step (calc_adaptive_rays), prune every 100
    R = size_batch(args.target_samples)
    for _ in range(args.pretrain):
        rays, gts = batch(R)
        trainer.step(rays, gts)
        R = common_rays(trainer.num_rays)
    R = size_batch(args.target_samples)                    # fixed for the timed region: per-GPU work is constant ("weak")

    def timed_steps(R, warmup, steps, timing_sink=None):
        """`warmup` untimed + exactly `steps` timed steps with a one-batch look-ahead (like a prefetching data loader: the
        next batch's occupancy test is issued early, so a step does not stall on its sample-count read-back).
        -> (seconds [max over ranks], packed samples of this rank, prunes that fell into the timed steps)"""
        rays, gts = batch(R)
        for _ in range(warmup):
            nrays, ngts = batch(R)
            trainer.step(rays, gts, prefetch=nrays)
            rays, gts = nrays, ngts
        C.TIMING = timing_sink
        if world > 1:
            dist.barrier()
        _sync()
        it0 = trainer.total_iterations
        t0 = time.perf_counter()
        samples = 0
        for _ in range(steps):
            nrays, ngts = batch(R)
            _, ns = trainer.step(rays, gts, prefetch=nrays)
            samples += ns
            rays, gts = nrays, ngts
        _sync()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if timing_sink is not None and hasattr(trainer, "native_timing_into"):
            trainer.native_timing_into(timing_sink)               # (the natively issued steps time their entry points themselves)
        C.TIMING = None
        prunes = sum(1 for it in range(it0, it0 + steps) if it > 1 and it % trainer.prune_every == 0)
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, samples, prunes

    def all_sum(v):
        if world > 1:
            t = torch.tensor([v], device=dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return int(t.item())
        return v

    timing = {}                                           # HIP-event timing of the hot kernels, live, on the launch stream
    if world > 1 or trainer.force_allreduce:
        trainer.comm_timing = []                          # per-step events around the collectives and the optimizer (side stream)
    with _regime("headline", dev):
        elapsed, total_samples, prunes_in = timed_steps(R, args.warmup, args.steps, timing)
    comm = trainer.comm_summary()
    trainer.comm_timing = None
    fused_elems_headline = int(getattr(trainer, "fused_elements_last", 0))
    # scratch of the binned hash-grid backward in this regime (record slots sized from what earlier launches filled)
    fits = [f.last for f in getattr(C, "_slot_fits", {}).values() if f.last]
    scratch = None
    if fits:
        last = max(fits, key=lambda f: f["workspace_bytes"])
        # per level: records written, slots (buckets x emitting workgroups), their capacity, the fullest slot.  A slot's fill is
        # lumpy (a ray's records of one level go to few buckets: consecutive cells along x are consecutive table entries, hashed
        # levels included - the hash's x prime is 1), so the fullest slot is ~3x the mean and the scratch ~4.5x the records.
        scratch = {"workspace_bytes": last["workspace_bytes"], "record_bytes_written": 8 * sum(last["records"]),
                   "per_level_scale": [round(x, 3) for x in last["scale"]],
                   "per_level_records": list(last["records"]), "per_level_slot_capacity": list(last["cap"]),
                   "per_level_fullest_slot": list(last["fill"])}
    if comm is None and selftest is not None:
        comm = {"note": "no per-phase events were recorded for the collectives in this run"}
    if comm is not None:
        comm["optimizer_path"] = "sharded (reduce-scatter + own slice + all-gather)" if getattr(trainer, "sharded_optimizer", False) \
            else "all-reduce + replicated update"
        comm["selftest"] = selftest
        comm["grad_bytes_on_the_wire_per_step"] = 4 * min(trainer._live_grad_numel(), trainer.flat.grad.numel())
    total_samples_all = all_sum(total_samples)
    cells_now = _leaf_cells(pipe)
    if args.pmc_child:
        if dist.is_initialized():
            dist.destroy_process_group()
        return None

    # ---- large_batch_regime: 8 x the reference's batch (the headline until round 5), learning rates x sqrt(batch ratio), same model
    large = None
    if args.large_target_samples > 0:
        R_large = size_batch(args.large_target_samples)
        base_lr, trainer.lr = trainer.lr, trainer.lr * large_lr
        timing_large = {}
        with _regime("large_batch_regime", dev):
            lg_elapsed, lg_samples, lg_prunes = timed_steps(R_large, min(args.warmup, 3), args.steps, timing_large)
        trainer.lr = base_lr
        large = dict(R=R_large, elapsed=lg_elapsed, samples=all_sum(lg_samples), prunes=lg_prunes, timing=timing_large,
                     fused_elems=int(getattr(trainer, "fused_elements_last", 0)))
        trainer.target_sample_size = args.target_samples

    # ---- the unchanged-application regime: the reference trainer's own step (multiview_trainer.py:111-180 through
    # BaseTrainer.iterate, base_trainer.py:316-342) over a copy of the same model state: fp16 autocast + GradScaler, autograd over
    # the modular pipeline, torch.optim.AdamW with the reference's parameter groups, SampleRays batches, two .item() per step
    trainer.sync_master()               # (sharded optimizer: the copy below must not start from rows that are stale on this rank)
    with _regime("dropin_regime", dev):
        dropin = dropin_regime(pipe, bank_o, bank_d, bank_rgb, args, world, dev) if args.dropin_steps > 0 else None

    # ---- one prune, timed on its own (it falls into the timed steps only every 100th iteration)
    _sync()
    tp = time.perf_counter()
    trainer.prune()
    _sync()
    prune_ms = 1e3 * (time.perf_counter() - tp)

    # ---- per-kernel roofline from the live HIP events (algorithmic bytes: SURVEY.md 8d / DESIGN.md)
    d = getattr(trainer, "_direct", None)
    hash_direct = d is not None and getattr(d, "hash_fast", False) and not getattr(trainer, "_last_step_modular", True)
    live_levels = min(NGP["num_lods"], d.zero_from_col // NGP["feature_dim"]) if hash_direct else NGP["num_lods"]
    work = work_table(amp, args.hidden, NGP["num_lods"], live_levels)
    peaks = {"hbm": (HBM_PEAK_GBS, "GB/s"), "mfma": (2500.0 if amp else 157.3, "TFLOP/s")}

    def roofline_of(timing, fused_elems):
        """roofline object of one regime from its HIP-event sink {entry point: [(start, end, units)]}.
        One GPU: the grid's AdamW step runs inside the hash-grid backward's reduce kernel (MultiviewTrainStep._fused_update_args), so
        that launch also does the optimizer's work for the `fused_elems` elements it updates: parameter and both moments read and
        written (24 B) + the bf16 copy (2 B); the gradient itself never leaves LDS.  That is a DIFFERENT quantity from SURVEY 8(d)'s
        backward bytes: `achieved` / `frac` are the backward's algorithmic bytes alone; the variant with the optimizer's bytes has
        its own key."""
        kern = {}
        for name, evs in timing.items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            units = [u for _, _, u in evs]
            kern[name] = dict(avg_ms=float(np.mean(ms)), launches=len(ms), avg_units=float(np.mean(units)), total_ms=float(np.sum(ms)))
        fused_opt_bytes = fused_elems * (24 + (2 if amp else 0))

        def rate(name, v):
            bound, per = work[name]
            r = per * v["avg_units"] / (v["avg_ms"] * 1e-3)
            return bound, (r / 1e9 if bound == "hbm" else r / 1e12)

        kern = {n: v for n, v in kern.items() if n in work}
        dominant = max(kern, key=lambda k: kern[k]["total_ms"]) if kern else None
        roofline = None
        if dominant:
            k = kern[dominant]
            bound, achieved = rate(dominant, k)
            peak, unit = peaks[bound]
            roofline = dict(bound=bound, kernel=dominant, achieved=achieved, peak=peak, unit=unit, frac=achieved / peak,
                            traffic=None, traffic_note="not measured (--no-pmc, multi-GPU run or profiler unavailable); see profiles/",
                            avg_launch_ms=k["avg_ms"], units_per_launch=k["avg_units"], work_per_unit=work[dominant][1],
                            levels_processed=live_levels, levels_of_the_table=NGP["num_lods"],
                            work_note="algorithmic bytes of the levels the kernels process: lod_idx = 15 zeroes the finest level's columns "
                                      "(hash_grid.py:226-229), so 15 of 16 levels are gathered / scattered; SURVEY 8(d)'s nominal 16-level "
                                      f"figures would be {work_table(amp, args.hidden)['hashgrid_fwd'][1]} / "
                                      f"{work_table(amp, args.hidden)['hashgrid_bwd'][1]} B per sample",
                            all_kernels={n: dict(avg_ms=v["avg_ms"], bound=rate(n, v)[0], achieved=rate(n, v)[1],
                                                 frac=rate(n, v)[1] / peaks[rate(n, v)[0]][0], work_per_unit=work[n][1])
                                         for n, v in kern.items()})
            if fused_opt_bytes and "hashgrid_bwd" in roofline["all_kernels"]:
                k = kern["hashgrid_bwd"]
                both = (work["hashgrid_bwd"][1] * k["avg_units"] + fused_opt_bytes) / (k["avg_ms"] * 1e-3) / 1e9
                roofline["all_kernels"]["hashgrid_bwd"]["with_fused_optimizer"] = {
                    "elements_updated_in_the_launch": fused_elems, "optimizer_bytes_per_launch": fused_opt_bytes,
                    "achieved": both, "frac": both / HBM_PEAK_GBS,
                    "note": "NOT the headline figure: backward bytes + what torch.optim.AdamW's step moves for the table rows the reduce "
                            "workgroups own (parameter + two moments read and written, bf16 copy written: 26 B per element); the "
                            "separate optimizer launch only covers the coarse levels, the frozen finest level and the decoder"}
                if dominant == "hashgrid_bwd":
                    roofline["with_fused_optimizer"] = roofline["all_kernels"]["hashgrid_bwd"]["with_fused_optimizer"]
            if "hashgrid_fwd" in roofline["all_kernels"]:
                # SURVEY 8(d)'s algorithmic bytes (588 B/sample, 512 of them gathered table entries) over the launch time exceed the HBM
                # peak: the 20.9 MB of tables are re-read from L2 / Infinity Cache, so that figure is NOT an HBM fraction and is kept
                # only under its own name.  What bounds the kernel is the L2-miss fill path (half of its L2 requests miss a 4 MB L2
                # holding a 20.9 MB table); its measured memory-side traffic replaces `achieved` / `frac` below when the PMC pass ran.
                e = roofline["all_kernels"]["hashgrid_fwd"]
                e["algorithmic_model"] = {"achieved": e["achieved"], "frac_of_hbm_peak": e["frac"],
                                          "note": "SURVEY 8(d) bytes / launch time; > 1 is possible because table gathers hit L2 / MALL"}
                e["bound"], e["achieved"], e["frac"] = "l2-miss-fill", None, None
                e["note"] = "memory-side traffic of the L2s (FETCH_SIZE x2 + WRITE_SIZE) over launch time against the 8 TB/s peak; not measured in this run"
        return roofline

    roofline = roofline_of(timing, fused_elems_headline)
    out = None
    if rank == 0:
        # quality: PSNR on held-out rays after pretrain + warm-up + both timed loops (real optimisation steps all of them)
        psnr = None
        if args.eval_rays > 0:
            with torch.no_grad(), torch.autocast(dev.type, dtype=torch.bfloat16, enabled=amp):
                eo, ed, ergb = synlego.ray_bank(args.eval_rays, seed=7, device=dev)
                chunks = []
                for s in range(0, eo.shape[0], 8192):
                    rb = pipe(rays=Rays(eo[s:s + 8192], ed[s:s + 8192], dist_min=synlego.NEAR, dist_max=synlego.FAR), channels=["rgb"])
                    chunks.append(rb.rgb.float())
                mse = float(((torch.cat(chunks) - ergb) ** 2).mean())
            psnr = 10 * math.log10(1.0 / max(mse, 1e-12))
        rays_total = R * args.steps * world
        # "prune amortised".  A window of K steps holds K/100 prunes on average but floor or ceil of that in fact
        # (none at all in the driver's 20 steps): `value` and `ms_per_step` always charge exactly K/100 prunes, at the cost of
        # the prune timed on its own below; the raw window is kept under `timed_window`.
        amort = elapsed + (args.steps / trainer.prune_every - prunes_in) * prune_ms * 1e-3 if trainer.prune_every > 0 else elapsed
        large_line = None
        if large is not None:
            lg_rays = large["R"] * args.steps * world
            lg_amort = large["elapsed"] + (args.steps / trainer.prune_every - large["prunes"]) * prune_ms * 1e-3 \
                if trainer.prune_every > 0 else large["elapsed"]
            large_line = {"target_samples_per_step": args.large_target_samples, "rays_per_step_per_gpu": large["R"],
                          "lr_scale": large_lr, "value": lg_rays / lg_amort, "unit": "rays/s", "ms_per_step": 1e3 * lg_amort / args.steps,
                          "timed_window": {"seconds": large["elapsed"], "ms_per_step": 1e3 * large["elapsed"] / args.steps,
                                           "value": lg_rays / large["elapsed"], "prunes_inside": large["prunes"]},
                          "samples_per_sec": large["samples"] / large["elapsed"],
                          "roofline": roofline_of(large["timing"], large["fused_elems"]),
                          "note": "8 x the reference trainer's batch (the headline `value` until round 5), learning rates x sqrt(batch "
                                  "ratio): more rays per second, each worth less - `quality` and profiles/r06_time_to_psnr_*.txt hold "
                                  "the held-out PSNR of both regimes at equal rays and at equal training seconds"}
        out = {
            "metric": "training rays/sec, HashGrid NeRF (nerf_hash.yaml), synthetic Lego 800x800",
            "value": rays_total / amort, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * amort / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "timed_window": {"seconds": elapsed, "ms_per_step": 1e3 * elapsed / args.steps, "value": rays_total / elapsed,
                             "prunes_inside": prunes_in, "prunes_charged": args.steps / trainer.prune_every if trainer.prune_every > 0 else 0.0,
                             "note": "`value` = rays / (window seconds + (steps / prune_every - prunes_inside) x prune.ms)"},
            "dtype": "bf16" if amp else "f32", "data": "synthetic",
            "config": {"workload": "app/nerf nerf_hash.yaml: OctreeAS level 7, HashGrid L=16 F=2 T=2^19 res 16..512 'cat', "
                                   f"NeRF hidden {args.hidden}, 'ray' raymarch {args.num_steps} candidates/ray, huber, AdamW; "
                                   f"SynLego 800x800 rays; occupancy: {args.occupancy} level-7 start, pruned every 100 steps "
                                   f"({args.pretrain} untimed pre-training steps before the timed ones)",
                       "rays_per_step_per_gpu": R, "target_samples_per_step": args.target_samples,
                       "batch_note": "the reference trainer's own batch: target_sample_size = 2**18 packed samples per step "
                                     "(multiview_trainer.py:58,95-109); until round 5 `value` was quoted at 8 x that (now "
                                     "large_batch_regime) - at equal rays the large batch ends 1.0 dB (lr x 3) to 3.6 dB (lr x 1) lower, "
                                     "profiles/r06_time_to_psnr_equal_rays.txt",
                       "global_target_samples_per_step": args.target_samples * world,
                       "scaling_note": ("strong: the global batch of --target-samples = %d packed samples is split over the ranks"
                                        % global_target) if args.scaling == "strong" else
                                       "weak: every GPU takes --target-samples packed samples per step (global batch grows with N)",
                       "samples_per_ray": total_samples_all / max(rays_total, 1), "parallelism": f"ray-sharded dp{world}",
                       "occupied_cells": cells_now, "true_occupied_cells": int(true_cells.shape[0]),
                       "prunes_inside_timed_steps": prunes_in},
            "samples_per_sec": total_samples_all / elapsed,
            "host_binding": host_binding or "none (no NUMA topology to bind to, or WISP_NUMA_BIND=0)",
            "prune": {"ms": prune_ms, "every_steps": trainer.prune_every, "value_without_any_prune":
                      rays_total / max(elapsed - prunes_in * prune_ms * 1e-3, 1e-9)},
            "large_batch_regime": large_line,
            "dropin_regime": dropin, "comm": comm,
            "psnr_db": psnr, "optimisation_steps_before_psnr": trainer.total_iterations,
            "roofline": roofline,
        }
        if roofline and scratch:
            roofline["hashgrid_bwd_scratch"] = scratch
        if world == 1 and roofline and not args.no_pmc:
            groups = {roofline["kernel"]: PMC_KERNELS[roofline["kernel"]]}
            if "hashgrid_fwd" in roofline["all_kernels"]:
                groups["hashgrid_fwd"] = PMC_KERNELS["hashgrid_fwd"]
            pmc, note = live_pmc_traffic(args, groups)
            roofline["traffic_note"] = note if pmc is not None else note
            if pmc is not None:
                roofline["traffic"] = pmc[roofline["kernel"]]["bytes"]
                roofline["traffic_detail"] = pmc[roofline["kernel"]]
                if "hashgrid_fwd" in pmc and pmc["hashgrid_fwd"]["bytes"] > 0:
                    e, t = roofline["all_kernels"]["hashgrid_fwd"], pmc["hashgrid_fwd"]
                    gbs = t["bytes"] / (e["avg_ms"] * 1e-3) / 1e9
                    req = t["l2_hit"] + t["l2_miss"]
                    e.update(achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, traffic=t["bytes"],
                             l2_requests_per_launch=req, l2_hit_rate=(t["l2_hit"] / req if req else None),
                             l2_request_gbs=(req * 128 / (e["avg_ms"] * 1e-3) / 1e9 if req else None), l2_peak_gbs=34500.0,
                             note="bound by the L2-miss fill path: `achieved` = memory-side bytes of the L2s per launch (live FETCH_SIZE x2 + "
                                  "WRITE_SIZE; Infinity-Cache hits included) / launch time, against the 8 TB/s HBM peak (6.29 TB/s is the "
                                  "copy rate the guide measures); l2_request_gbs = L2 requests x 128 B against the L2s' 34.5 TB/s")
        if world == 1 and not args.no_configs:
            # the other BASELINE.json configurations + the reference's best published row (hidden 128), a few steps each
            import bench_configs
            out["configs"] = bench_configs.secondary_lines(args, dev)
            out["configs"]["hidden128"] = hidden128_line(args, dev, pipe, batch, args.target_samples)
            if args.large_target_samples > 0 and "error" not in out["configs"]["hidden128"]:
                big = hidden128_line(args, dev, pipe, batch, args.large_target_samples)
                out["configs"]["hidden128"]["large_batch_regime"] = big
        if world == 1 and args.quality_budget > 0:
            out["quality"] = quality_table(args, dev, (bank_o, bank_d, bank_rgb), amp, large_lr)
        if world == 1 and args.dp_steps > 0 and not dist.is_initialized():
            out["dp_path_regime"] = dp_path_regime(args, dev, pipe, batch, R, out["ms_per_step"])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(true_cells, args.hidden, args.num_steps)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        guard.emit(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
