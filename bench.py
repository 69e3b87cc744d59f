#!/usr/bin/env python
"""bench.py - training rays/sec of HashGrid NeRF (app/nerf nerf_hash.yaml shape) on the synthetic Lego stand-in.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of rays: on-device ray sampling from the bank, raymarch
('ray', 2048 candidates per ray against the level-7 occupancy octree), hash-grid interpolation (L=16, F=2, T=2^19),
density + colour MLPs, packed compositing, huber loss, backward, (RCCL all-reduce when N>1), fused AdamW.
Prints ONE JSON line (rank 0).  Inputs are synthetic (synlego.py) and resident in HBM before the timed region.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--num-steps", type=int, default=2048, help="raymarch candidates per ray (nerf_hash.yaml:49)")
    ap.add_argument("--target-samples", type=int, default=2 ** 21,
                    help="packed samples per step per GPU (reference trainer default is 2^18)")
    ap.add_argument("--bank-rays", type=int, default=2 ** 21)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval-rays", type=int, default=2 ** 16)
    return ap.parse_args()


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


def cpu_baseline(blas_cells, hidden, num_steps, budget_s=20.0):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores: same model shape, same
    occupancy, R = 256 rays per step, as many steps as fit the budget (at least 2)."""
    from oracle import nerf as onerf, spc as ospc
    import synlego
    torch.manual_seed(0)
    res = [int(np.floor(16 * (np.exp((np.log(512) - np.log(16)) / 15) ** l))) for l in range(16)]
    nef = onerf.OracleNeRF(res, 2, 19, 'cat', 1e-9, hidden, 1, True, 4)
    blas = onerf.OracleBLAS(ospc.points_to_octree(blas_cells.cpu().numpy(), 7))
    opt = onerf.make_optimizer(nef)
    R = 256
    o, d, _ = synlego.ray_bank(R, seed=123, device='cpu', with_gt=False)
    gts = torch.rand(R, 3)
    rng = np.random.default_rng(0)
    onerf.train_step(nef, blas, opt, o, d, gts, 1.0, 5.0, num_steps, rng.uniform(size=(R, num_steps)).astype(np.float32))
    t0, n, samples = time.time(), 0, 0
    while n < 2 or (time.time() - t0) < budget_s:
        _, s = onerf.train_step(nef, blas, opt, o, d, gts, 1.0, 5.0, num_steps,
                                rng.uniform(size=(R, num_steps)).astype(np.float32))
        n += 1
        samples += s
        if n >= 40:
            break
    dt = time.time() - t0
    return dict(value=R * n / dt, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} oracle train steps of {R} rays x {num_steps} candidates ({samples // max(n,1)} packed samples/step), "
                       f"fp32, torch-CPU + numpy, {dt:.1f}s")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ       # started by torch.distributed.run
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)          # nccl == RCCL on ROCm
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import synlego
    import wisp._C as C
    from wisp.core import Rays
    from wisp.trainers import MultiviewTrainStep

    cells = synlego.occupied_cells(7, device=dev)
    pipe = build_pipeline(dev, args.hidden, args.num_steps, cells)
    amp = args.precision == "bf16"
    trainer = MultiviewTrainStep(pipe, lr=1e-3, eps=1e-16, weight_decay=1e-6, grid_lr_weight=500.0, rgb_loss_type='huber',
                                 prune_every=100, target_sample_size=args.target_samples, max_rays=2 ** 18, enable_amp=amp)

    # ---- ray bank resident in HBM (this rank's shard of the training rays) + ground truth
    bank_o, bank_d, bank_rgb = synlego.ray_bank(args.bank_rays, seed=1000 + rank, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    def batch(n):
        idx = torch.randint(0, bank_o.shape[0], (n,), device=dev, generator=gen)
        o, d, rgb = C.gather_rows(idx, [bank_o, bank_d, bank_rgb])       # SampleRays: one launch for the three gathers
        return Rays(o, d, dist_min=synlego.NEAR, dist_max=synlego.FAR), rgb

    # warm-up raymarch sizes the batch like MultiviewTrainer.step's first call (multiview_trainer.py:119-122)
    rays, _ = batch(4096)
    rm = pipe.nef.grid.raymarch(rays, level=pipe.nef.grid.active_lods[-1], num_samples=args.num_steps, raymarch_type='ray')
    pipe.tracer.prev_num_samples = rm.samples.shape[0]
    R = trainer.calc_adaptive_rays(4096)
    if world > 1:                                         # every rank uses the same ray count
        t = torch.tensor([R], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        R = int(t.item())

    # the loop hands the trainer the NEXT batch's rays as well (a one-batch look-ahead, like a prefetching data loader):
    # their occupancy test is issued early and the next step does not stall on its sample-count read-back
    rays, gts = batch(R)
    for _ in range(args.warmup):
        nrays, ngts = batch(R)
        trainer.step(rays, gts, prefetch=nrays)
        rays, gts = nrays, ngts

    C.TIMING = {}                                         # HIP-event timing of the hash-grid kernels, live
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total_samples = 0
    for _ in range(args.steps):
        nrays, ngts = batch(R)
        _, ns = trainer.step(rays, gts, prefetch=nrays)
        total_samples += ns
        rays, gts = nrays, ngts
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timing, C.TIMING = C.TIMING, None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([total_samples], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_samples_all = int(t.item())
    else:
        total_samples_all = total_samples

    # ---- per-kernel roofline from the live HIP events (algorithmic bytes: SURVEY.md 8d / DESIGN.md)
    kern = {}
    for name, evs in (timing or {}).items():
        ms = [a.elapsed_time(b) for a, b, _ in evs]
        units = [u for _, _, u in evs]
        kern[name] = dict(avg_ms=float(np.mean(ms)), launches=len(ms), avg_units=float(np.mean(units)), total_ms=float(np.sum(ms)))
    b = 2 if amp else 4
    # algorithmic work per packed sample (SURVEY.md 8d / DESIGN.md): bytes for the HBM-bound kernels, flops for the MLP
    work = {"hashgrid_fwd": ("hbm", 12 + 16 * 8 * 2 * b + 16 * 2 * b),        # coords + 128 gathered entries + 32 outputs
            # 12 + L*F*b + 2*L*2^d*F*b_acc with b_acc = the table element size (1100 B for 16-bit tables).
            # The kernels accumulate in fp32 / 64-bit fixed point and merge runs, so what they actually move is `traffic`.
            "hashgrid_bwd": ("hbm", 12 + 16 * 2 * b + 2 * 16 * 8 * 2 * b),
            "nerf_mlp_fwd": ("mfma", 20096), "nerf_mlp_bwd": ("mfma", 3 * 20096)}
    peaks = {"hbm": (HBM_PEAK_GBS, "GB/s"), "mfma": (2500.0 if amp else 157.3, "TFLOP/s")}

    def rate(name, v):
        bound, per = work[name]
        r = per * v["avg_units"] / (v["avg_ms"] * 1e-3)
        return bound, (r / 1e9 if bound == "hbm" else r / 1e12)

    def pmc_traffic(kernel_prefixes):
        """HBM-side bytes per launch from the committed rocprofv3 PMC summaries of this same command (profiles/r01_pmc_*):
        FETCH_SIZE (KiB, doubled per MI355X_MICROARCH.md: it counts 128-B requests as 64 B) + WRITE_SIZE (KiB)."""
        import csv
        tot = 0.0
        try:
            for fname, mult in (("r01_pmc_FETCH_SIZE.csv", 2.0), ("r01_pmc_WRITE_SIZE.csv", 1.0)):
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    for row in csv.DictReader(f):
                        if any(k in row["kernel"] for k in kernel_prefixes):
                            tot += float(row["mean_per_dispatch"]) * 1024.0 * mult
        except (OSError, KeyError, ValueError):
            return None
        return tot or None

    pmc_names = {"hashgrid_fwd": ["hashgrid_fwd_kernel"], "hashgrid_bwd": ["hashgrid_bwd_emit_kernel", "hashgrid_bwd_reduce_kernel",
                                                                           "hashgrid_bwd_kernel"],
                 "nerf_mlp_fwd": ["mlp_fwd_kernel"], "nerf_mlp_bwd": ["mlp_bwd_kernel", "nerf_mlp_reduce_kernel"]}
    kern = {n: v for n, v in kern.items() if n in work}
    dominant = max(kern, key=lambda k: kern[k]["total_ms"]) if kern else None
    roofline = None
    if dominant:
        k = kern[dominant]
        bound, achieved = rate(dominant, k)
        peak, unit = peaks[bound]
        roofline = dict(bound=bound, kernel=dominant, achieved=achieved, peak=peak, unit=unit, frac=achieved / peak,
                        traffic=pmc_traffic(pmc_names[dominant]) if (amp and world == 1) else None,
                        traffic_note="bytes/launch from profiles/r01_pmc_{FETCH,WRITE}_SIZE.csv (rocprofv3 --pmc, same command, "
                                     "FETCH_SIZE x2); hashgrid_bwd = emit + reduce kernels",
                        avg_launch_ms=k["avg_ms"], units_per_launch=k["avg_units"],
                        work_per_unit=work[dominant][1],
                        all_kernels={n: dict(avg_ms=v["avg_ms"], bound=rate(n, v)[0], achieved=rate(n, v)[1],
                                             frac=rate(n, v)[1] / peaks[rate(n, v)[0]][0]) for n, v in kern.items()})

    out = None
    if rank == 0:
        # quality probe: PSNR on held-out rays after the (short) run
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            eo, ed, ergb = synlego.ray_bank(args.eval_rays, seed=7, device=dev)
            chunks = []
            for s in range(0, eo.shape[0], 8192):
                rb = pipe(rays=Rays(eo[s:s + 8192], ed[s:s + 8192], dist_min=synlego.NEAR, dist_max=synlego.FAR), channels=["rgb"])
                chunks.append(rb.rgb.float())
            mse = float(((torch.cat(chunks) - ergb) ** 2).mean())
        psnr = 10 * math.log10(1.0 / max(mse, 1e-12))
        rays_total = R * args.steps * world
        out = {
            "metric": "training rays/sec, HashGrid NeRF (nerf_hash.yaml), synthetic Lego 800x800",
            "value": rays_total / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if amp else "f32", "data": "synthetic",
            "config": {"workload": "app/nerf nerf_hash.yaml: OctreeAS level 7, HashGrid L=16 F=2 T=2^19 res 16..512 'cat', "
                                   f"NeRF hidden {args.hidden}, 'ray' raymarch {args.num_steps} candidates/ray, huber, AdamW; "
                                   "SynLego 800x800 rays, analytic (post-prune) occupancy",
                       "rays_per_step_per_gpu": R, "target_samples_per_step": args.target_samples,
                       "samples_per_ray": total_samples_all / max(rays_total, 1), "parallelism": f"ray-sharded dp{world}",
                       "occupied_cells": int(cells.shape[0])},
            "samples_per_sec": total_samples_all / elapsed,
            "psnr_db_after_run": psnr,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cells, args.hidden, args.num_steps)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
