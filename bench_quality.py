"""Time-to-quality of the nerf_hash.yaml training step at different batch sizes (VERDICT r5 missing-2 / next-3).

BASELINE.json's metric is "training rays/sec + PSNR"; the reference quotes PSNR at FIXED RAY BUDGETS (epochs 100 / 200 / 300 of
100 views x 4096 rays, docs/pages/app_nerf.md:185-200; one epoch = base_trainer.py:198-203).  The headline regime of bench.py packs
2^21 samples per step - 8 x the reference trainer's `target_sample_size = 2**18` (multiview_trainer.py:58) - so a ray budget buys
8 x fewer optimizer steps there, and an 8-GPU weak-scaling run another 8 x fewer.  This module trains the SAME model from the
SAME dense level-7 start on the SAME ray stream under each regime and logs held-out PSNR against rays consumed and against
training wall-clock (evaluation excluded):

  ref_2p18         target 2^18 packed samples per step              (the reference trainer's batch)
  headline_2p21    target 2^21                                        (bench.py's `value`)
  dp8_8x2p21       8 micro-batches of 2^21 per optimizer step         (MultiviewTrainStep.accumulate: the arithmetic of 8 data-parallel
                                                                      ranks, each with the headline's per-GPU batch - weak scaling)
  + any of them with a learning-rate rule `lr_scale` (a multiplier on both parameter groups' learning rates)

`time_to_psnr` is imported by bench.py (a short form: `psnr_at_rays` in the JSON line) and driven over several seeds by
scripts/time_to_psnr.py (profiles/r06_time_to_psnr.txt)."""
import math
import time

import torch

REFERENCE_RAY_BUDGET = 100 * 100 * 4096          # 100 epochs x 100 views x 4096 rays (nerf_hash.yaml; base_trainer.py:198-203)
REGIMES = {
    "ref_2p18": dict(target=2 ** 18, accum=1),
    "mid_2p19": dict(target=2 ** 19, accum=1),
    "mid_2p20": dict(target=2 ** 20, accum=1),
    "headline_2p21": dict(target=2 ** 21, accum=1),
    "dp8_8x2p21": dict(target=2 ** 21, accum=8),
}


def _psnr(pipe, bank, amp, Rays, near, far):
    eo, ed, ergb = bank
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        pipe.eval()
        se = 0.0
        for s in range(0, eo.shape[0], 8192):
            rb = pipe(rays=Rays(eo[s:s + 8192], ed[s:s + 8192], dist_min=near, dist_max=far), channels=["rgb"])
            se += float(((rb.rgb.float() - ergb[s:s + 8192]) ** 2).sum())
        pipe.train()
    return 10 * math.log10(1.0 / max(se / ergb.numel(), 1e-12))


def time_to_psnr(dev, regime, train_bank, eval_bank, ray_budget=REFERENCE_RAY_BUDGET, checkpoints=(1e7, 2e7, 4e7), seed=0,
                 hidden=64, num_steps=2048, amp=True, lr_scale=1.0, log_every_rays=None, max_rays=2 ** 18, scheduler=True,
                 prune_every=100):
    """Train one fresh nerf_hash.yaml model under `regime` (a REGIMES key or a dict(target=, accum=)) until `ray_budget` rays have
    been consumed.  scheduler: nerf_hash.yaml's MultiStepLR (x 0.333 at 0.5 / 0.75 / 0.9 of the run, base_trainer.py:238-246), placed at
    those fractions of the RAY budget so that every regime decays at the same point of its data.  -> dict(psnr_at_rays={rays: dB}, curve=[(rays, steps, train seconds, dB)], ...)."""
    import bench
    import synlego
    import wisp._C as C
    from wisp.accelstructs import OctreeAS
    from wisp.core import Rays
    from wisp.trainers import MultiviewTrainStep
    cfg = REGIMES[regime] if isinstance(regime, str) else dict(regime)
    target, accum = int(cfg["target"]), int(cfg.get("accum", 1))
    cells = OctreeAS.make_dense(level=7).points[-(128 ** 3):].to(dev)            # nerf_hash.yaml:16-17
    pipe = bench.build_pipeline(dev, hidden, num_steps, cells)                    # (seeds torch with 0: same initial weights)
    tr = MultiviewTrainStep(pipe, lr=1e-3 * lr_scale, eps=1e-16, weight_decay=1e-6, grid_lr_weight=500.0, rgb_loss_type='huber',
                            prune_every=prune_every, target_sample_size=target, max_rays=max_rays, enable_amp=amp, seed=seed)
    tr.grad_accum_steps = accum
    bank_o, bank_d, bank_rgb = train_bank
    gen = torch.Generator(device=dev).manual_seed(4321 + seed)

    def batch(n):
        idx = torch.randint(0, bank_o.shape[0], (n,), device=dev, generator=gen)
        o, d, rgb = C.gather_rows(idx, [bank_o, bank_d, bank_rgb])
        return Rays(o, d, dist_min=synlego.NEAR, dist_max=synlego.FAR), rgb

    marks = sorted(set(float(c) for c in checkpoints if c <= ray_budget))
    if log_every_rays:
        marks = sorted(set(marks) | {float(k * log_every_rays) for k in range(1, int(ray_budget // log_every_rays) + 1)})
    curve, at = [], {}
    decay_at = [f * ray_budget for f in (0.5, 0.75, 0.9)] if scheduler else []
    R, rays_done, steps, train_s = 4096, 0, 0, 0.0
    nxt = 0
    torch.cuda.synchronize()
    while rays_done < ray_budget:
        t0 = time.perf_counter()
        for _ in range(accum - 1):
            rays, gts = batch(R)
            tr.accumulate(rays, gts)
            rays_done += R
        rays, gts = batch(R)
        tr.step(rays, gts)
        rays_done += R
        steps += 1
        while decay_at and rays_done >= decay_at[0]:
            decay_at.pop(0)
            tr.milestones = sorted(tr.milestones + [tr.opt_steps])               # MultiStepLR milestone = "from the next step on"
        R = max(256, tr.num_rays)                                                # calc_adaptive_rays, every step
        if nxt < len(marks) and rays_done >= marks[nxt]:
            torch.cuda.synchronize()
            train_s += time.perf_counter() - t0
            tr.wait_for_parameters()
            db = _psnr(pipe, eval_bank, amp, Rays, synlego.NEAR, synlego.FAR)
            while nxt < len(marks) and rays_done >= marks[nxt]:
                if marks[nxt] in [float(c) for c in checkpoints]:
                    at[int(marks[nxt])] = db
                nxt += 1
            curve.append((rays_done, steps, train_s, db))
            torch.cuda.synchronize()
        else:
            train_s += time.perf_counter() - t0                                  # (asynchronous issue: exact at the synchronised marks)
    torch.cuda.synchronize()
    blas = pipe.nef.grid.blas
    return dict(regime=regime if isinstance(regime, str) else "custom", target_samples_per_step=target, micro_batches_per_step=accum,
                lr_scale=lr_scale, prune_every=prune_every, seed=seed, rays=rays_done, optimizer_steps=steps, train_seconds=train_s,
                rays_per_step_at_end=R * accum, leaf_cells_at_end=int(blas.pyramid[0, blas.max_level]),
                psnr_at_rays={str(k): v for k, v in at.items()}, curve=curve)
