"""bench.py --config {v8, vqad, nglod}: the other BASELINE.json configurations on their synthetic stand-ins.

  v8     C4, one GPU's share: HashGrid NeRF (nerf_hash.yaml model) over `OctreeAS.from_pointcloud(level 7)` of the SynV8 depth
         point cloud, 'voxel' march with 16 samples per intersected cell, white background, mip-2 (400x400) rays
         (tests/apps/test_nerf.py:65-87).
  vqad   C5: CodebookOctreeGrid F=5, 4 LODs on a level-8 octree from the point cloud, 4-bit codebooks, decoders without bias,
         'voxel' N=16, white background, RMSprop lr 1e-3, grid lr x100, L2 loss (app/nerf/configs/nerf_codebook.yaml).
  nglod  C3: OctreeGrid F=16, 6 LODs on a level-7 octree from SynArmadillo surface samples, NeuralSDF 19 -> 128 -> 1,
         Adam, 512 coordinates per step (app/nglod/configs/nglod_octree.yaml) + sphere-traced rendering (32 steps x 0.8).

Each prints ONE JSON line in bench.py's format (value = whole-job throughput with inputs resident in HBM) plus `kernels`:
every C-ABI launch of the timed steps, HIP-event timed on the launch stream, and a `roofline` object for the dominant one
when its algorithmic bytes are defined below.  These are secondary lines (profiles/), not the driver's headline."""
import json
import math
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


def _kernel_table(sink):
    out = {}
    for name, evs in sink.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[name.replace("wisp_", "")] = dict(launches=len(ms), avg_ms=float(np.mean(ms)), total_ms=float(np.sum(ms)))
    tot = sum(v["total_ms"] for v in out.values()) or 1.0
    for v in out.values():
        v["share"] = v["total_ms"] / tot
    return dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms"]))


def _busy_union_ms(sink):
    """(GPU-busy ms, span ms) of the event-bracketed launches: busy = the UNION of their intervals - launches on side streams
    (look-ahead raytrace counts, the optimizer) overlap the main stream's, and a sum of durations counts those stretches twice;
    span = first launch start .. last launch end of the same pass."""
    pairs = [p for evs in sink.values() for p in evs]
    if not pairs:
        return 0.0, 0.0
    ref = pairs[0][0]                                            # any event: only differences matter
    spans = sorted((ref.elapsed_time(a), ref.elapsed_time(b)) for a, b in pairs)
    busy, cur_s, cur_e = 0.0, spans[0][0], spans[0][1]
    for s0, e0 in spans[1:]:
        if s0 > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s0, e0
        else:
            cur_e = max(cur_e, e0)
    return busy + (cur_e - cur_s), max(e for _, e in spans) - spans[0][0]


# entry points whose time is NOT set by HBM bytes, with what sets it (measured: profiles/); their byte model
# is still reported, under `hbm_model`, because SURVEY 8(d) prescribes it
NOT_HBM_BOUND = {
    "codebook_trilinear_multi_bwd": (
        "valu+atomics",
        "tables of 0.2 M rows sit in L2 / Infinity Cache; the launch is the wave-level run merge (320 DPP / packed-fma instructions per "
        "sample and level, ~115 us of its ~220 us scatter at 2 M samples) plus ~3 M row-wide 64-bit atomic requests retired on the memory "
        "side (~95 us: the same kernel with the atomics compiled out, scripts/gpu_r4_c.sh), a magnitude pass (32 us) and the row pass (32 us)"),
}


def _roofline(kernels, bytes_per_launch, steps):
    """dominant C-ABI entry point of the step vs the HBM roofline, when its algorithmic bytes are known."""
    for name, v in kernels.items():
        if name in bytes_per_launch:
            achieved = bytes_per_launch[name] / (v["avg_ms"] * 1e-3) / 1e9
            model = dict(achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                         algorithmic_bytes_per_launch=bytes_per_launch[name])
            if name in NOT_HBM_BOUND:
                bound, why = NOT_HBM_BOUND[name]
                return dict(bound=bound, kernel=name, achieved=None, peak=None, unit=None, frac=None, traffic=None, avg_launch_ms=v["avg_ms"],
                            share_of_kernel_time=v["share"], hbm_model=model, note=why)
            return dict(bound="hbm", kernel=name, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                        traffic=None, avg_launch_ms=v["avg_ms"], algorithmic_bytes_per_launch=bytes_per_launch[name],
                        share_of_kernel_time=v["share"])
        if v["share"] > 0.05:
            return dict(bound=None, kernel=name, note="dominant launch has no algorithmic-byte model here (torch GEMM or host-driven)",
                        avg_launch_ms=v["avg_ms"], share_of_kernel_time=v["share"])
    return None


def _latency_roofline(kernels, bytes_per_launch, steps, replay_ms_per_step):
    """A 512-coordinate step moves ~1.7 MB per launch: no bandwidth or matrix roofline says anything about it.  What bounds it is
    the number of launches times the per-launch floor of the device (the shortest launch of the table is that floor: a
    kernel that does almost nothing still takes it).  `frac` = that floor x launches / the measured (graph-replayed) step."""
    per_step = {n: v["launches"] / steps for n, v in kernels.items()}
    launches = sum(per_step.values())
    floor_ms = min(v["avg_ms"] for v in kernels.values()) if kernels else None
    dom = next(iter(kernels)) if kernels else None
    out = dict(bound="latency", kernel=dom, launches_per_step=launches, per_launch_floor_ms=floor_ms,
               floor_ms_per_step=None if floor_ms is None else floor_ms * launches, measured_ms_per_step=replay_ms_per_step,
               frac=None if floor_ms is None else min(1.0, floor_ms * launches / replay_ms_per_step), unit="ms", traffic=None,
               launches_by_entry_point=per_step,
               note="frac = (launches x shortest launch) / measured step: how close the step is to pure launch cadence; the HBM model "
                    "of the same launches is in `hbm_model` for completeness")
    if dom in bytes_per_launch:
        v = kernels[dom]
        gbs = bytes_per_launch[dom] / (v["avg_ms"] * 1e-3) / 1e9
        out["hbm_model"] = dict(kernel=dom, achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                                algorithmic_bytes_per_launch=bytes_per_launch[dom])
    return out


def _nerf_run(args, dev, pipe, trainer, bank, steps, warmup, label, metric, bytes_fn, extra_bytes_fn=None, roofline_extra=None):
    import wisp._C as C
    from wisp.core import Rays
    import synlego
    bank_o, bank_d, bank_rgb = bank
    gen = torch.Generator(device=dev).manual_seed(1234)

    def batch(n):
        idx = torch.randint(0, bank_o.shape[0], (n,), device=dev, generator=gen)
        o, d, rgb = C.gather_rows(idx, [bank_o, bank_d, bank_rgb])
        return Rays(o, d, dist_min=synlego.NEAR, dist_max=synlego.FAR), rgb

    R = 4096
    for _ in range(args.pretrain):
        rays, gts = batch(R)
        trainer.step(rays, gts)
        R = max(256, trainer.num_rays)
    rays, gts = batch(R)
    for _ in range(warmup):
        nrays, ngts = batch(R)
        trainer.step(rays, gts, prefetch=nrays)
        rays, gts = nrays, ngts
    def loop(n):
        nonlocal rays, gts
        total = 0
        for _ in range(n):                # one-batch look-ahead, like bench.py's main loop (a prefetching data loader)
            nrays, ngts = batch(R)
            _, ns = trainer.step(rays, gts, prefetch=nrays)
            total += ns
            rays, gts = nrays, ngts
        return total

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = loop(steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # the per-launch table comes from a second pass: a pair of HIP events around EVERY launch costs the step a marker packet
    # before and after each kernel (measured: ~0.1 ms of a 2.4 ms step on the codebook line), so it must not sit in the timed one
    psteps = max(8, min(steps, 32))
    C.TIMING_ALL = {}
    loop(psteps)
    torch.cuda.synchronize()
    sink, C.TIMING_ALL = C.TIMING_ALL, None
    kernels = _kernel_table(sink)
    busy_ms, span_ms = _busy_union_ms(sink)
    S = samples / steps
    roofline = _roofline(kernels, bytes_fn(S, R), psteps)
    if roofline and roofline.get("bound") == "hbm":
        roofline.update(roofline_extra or {})
        extra = (extra_bytes_fn(S, R) if extra_bytes_fn else {}).get(roofline["kernel"], 0)
        if extra:
            both = (roofline["algorithmic_bytes_per_launch"] + extra) / (roofline["avg_launch_ms"] * 1e-3) / 1e9
            roofline["with_fused_optimizer"] = {
                "optimizer_bytes_per_launch": extra, "achieved": both, "frac": both / HBM_PEAK_GBS,
                "note": "NOT the roofline figure: backward bytes + what AdamW's step moves for the table rows the reduce workgroups "
                        "own (parameter + two moments read and written, bf16 copy written: 26 B per element)"}
    with torch.no_grad():
        eo, ed, ergb = bank_o[:16384], bank_d[:16384], bank_rgb[:16384]
        rb = pipe(rays=Rays(eo, ed, dist_min=synlego.NEAR, dist_max=synlego.FAR), channels=["rgb"])
        psnr = 10 * math.log10(1.0 / max(float(((rb.rgb.float() - ergb) ** 2).mean()), 1e-12))
    return {"metric": metric, "value": R * steps / elapsed, "unit": "rays/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if trainer.enable_amp else "f32", "data": "synthetic",
            "config": {"workload": label, "rays_per_step_per_gpu": R, "samples_per_ray": S / R, "samples_per_step": S,
                       "pretrain_steps": args.pretrain},
            "samples_per_sec": samples / elapsed, "psnr_db_train_rays": psnr,
            "gpu_busy_fraction": busy_ms / span_ms if span_ms > 0 else None,
            "gpu_busy_note": "union of the launch intervals of the event-timed pass / that pass's own span, first launch to last (<= 1 by "
                             "construction; the pair of event markers around every launch is inside the span, so this is a lower bound "
                             "of the untimed-events steps' busy share)",
            "roofline": roofline, "kernels": kernels}


V8_CLOUD_RAYS, V8_LEVEL = 1 << 20, 7            # C4: SynV8 depth point cloud -> OctreeAS.from_pointcloud(level 7)
VQAD_CLOUD_RAYS, VQAD_LEVEL = 1 << 21, 8        # C5: level-8 octree from the point cloud
NGLOD_SURFACE_POINTS, NGLOD_LEVEL = 1 << 20, 7  # C3: level-7 octree from SynArmadillo surface samples


def v8_scene(dev):
    """The C4 scene exactly as `--config v8` times it (tests/test_gpu_0_parity.py checks parity on THIS object graph):
    -> (cloud, blas, grid, nef, pipe)."""
    import synlego
    from wisp.accelstructs import OctreeAS
    from wisp.models import Pipeline
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    import bench
    torch.manual_seed(0)
    cloud = synlego.v8_pointcloud(V8_CLOUD_RAYS, res=400, device=dev)
    blas = OctreeAS.from_pointcloud(cloud, V8_LEVEL)
    grid = HashGrid.from_geometric(blas, **bench.NGP)
    nef = NeuralRadianceField(grid, pos_embedder='none', view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1,
                              bias=True, prune_density_decay=None, prune_min_density=None).to(dev)
    pipe = Pipeline(nef, PackedRFTracer(raymarch_type='voxel', num_steps=16, bg_color=(1.0, 1.0, 1.0)))
    return cloud, blas, grid, nef, pipe


def vqad_scene(dev):
    """The C5 scene exactly as `--config vqad` times it -> (cloud, blas, grid, nef, pipe)."""
    import synlego
    from wisp.accelstructs import OctreeAS
    from wisp.models import Pipeline
    from wisp.models.grids import CodebookOctreeGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    torch.manual_seed(0)
    cloud = synlego.v8_pointcloud(VQAD_CLOUD_RAYS, res=400, device=dev)
    blas = OctreeAS.from_pointcloud(cloud, VQAD_LEVEL)
    grid = CodebookOctreeGrid(blas, feature_dim=5, num_lods=4, interpolation_type='linear', multiscale_type='sum', feature_std=0.01,
                              feature_bias=0.0, codebook_bitwidth=4)
    nef = NeuralRadianceField(grid, pos_embedder='none', view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1,
                              bias=False, prune_density_decay=None, prune_min_density=None).to(dev)
    pipe = Pipeline(nef, PackedRFTracer(raymarch_type='voxel', num_steps=16, bg_color=(1.0, 1.0, 1.0)))
    return cloud, blas, grid, nef, pipe


def nglod_scene(dev):
    """The C3 scene exactly as `--config nglod` times it -> (surface points, blas, grid, nef)."""
    import synlego
    from wisp.accelstructs import OctreeAS
    from wisp.models.grids import OctreeGrid
    from wisp.models.nefs import NeuralSDF
    torch.manual_seed(0)
    surf = synlego.armadillo_surface_points(NGLOD_SURFACE_POINTS, device=dev)
    blas = OctreeAS.from_pointcloud(surf, NGLOD_LEVEL)
    grid = OctreeGrid(blas, feature_dim=16, num_lods=6, interpolation_type='linear', multiscale_type='sum', feature_std=0.01)
    nef = NeuralSDF(grid, pos_embedder='none', position_input=True, hidden_dim=128, num_layers=1).to(dev)
    return surf, blas, grid, nef


def run_v8(args, dev):
    import synlego
    import bench
    from wisp.trainers import MultiviewTrainStep
    _, blas, grid, nef, pipe = v8_scene(dev)
    tr = MultiviewTrainStep(pipe, lr=1e-3, eps=1e-16, weight_decay=1e-6, grid_lr_weight=500.0, rgb_loss_type='huber', prune_every=-1,
                            target_sample_size=args.target_samples, enable_amp=args.precision == "bf16")
    o, d, _ = synlego.ray_bank(1 << 20, res=400, seed=1000, device=dev, with_gt=False)
    bank = (o, d, synlego.render_gt_white(o, d))
    # Same bookkeeping as the headline (bench.py::work_table): bytes of the levels the kernels PROCESS - the tracer's lod_idx = 15
    # zeroes the finest level's columns, so 15 of 16 levels are gathered / scattered: 556 / 1036 B per sample, not SURVEY 8(d)'s
    # nominal 588 / 1100 - and the AdamW step folded into the reduce launch is a separate quantity under its own key
    # (`with_fused_optimizer`), never part of `achieved` / `frac`.
    live = grid.num_lods - 1 if grid.multiscale_type == 'cat' else grid.num_lods
    work = bench.work_table(tr.enable_amp, 64, grid.num_lods, live)
    fwd_b, bwd_b = work["hashgrid_fwd"][1], work["hashgrid_bwd"][1]
    bytes_fn = lambda S, R: {"hashgrid_interpolate_bwd": bwd_b * S, "hashgrid_interpolate_bwd_adamw": bwd_b * S,
                             "hashgrid_interpolate_fwd": fwd_b * S,
                             "raymarch_voxel_emit": 37 * S, "composite_fwd": 25 * S, "composite_bwd": 41 * S}
    extra_fn = lambda S, R: {"hashgrid_interpolate_bwd_adamw":
                             getattr(tr, "fused_elements_last", 0) * (24 + (2 if tr.enable_amp else 0))}
    return _nerf_run(args, dev, pipe, tr, bank, args.steps, args.warmup,
                     f"C4 (one GPU): nerf_hash model over from_pointcloud(level 7) of SynV8 ({int(blas.pyramid[0, 7])} cells), "
                     "'voxel' 16 samples per cell, white background, 400x400 rays, AdamW",
                     "training rays/sec, HashGrid NeRF, synthetic V8 (voxel march)", bytes_fn, extra_fn,
                     dict(levels_processed=live, levels_of_the_table=grid.num_lods, work_per_unit=bwd_b))


def run_vqad(args, dev):
    import synlego
    from wisp.trainers import MultiviewTrainStep
    _, blas, grid, nef, pipe = vqad_scene(dev)
    tr = MultiviewTrainStep(pipe, lr=1e-3, eps=1e-8, weight_decay=0.0, grid_lr_weight=100.0, rgb_loss_type='l2', prune_every=-1,
                            target_sample_size=args.target_samples, enable_amp=args.precision == "bf16", optimizer='rmsprop')
    o, d, _ = synlego.ray_bank(1 << 20, res=400, seed=1000, device=dev, with_gt=False)
    bank = (o, d, synlego.render_gt_white(o, d))
    L = 4
    rows = [int(f.shape[0]) for f in grid.features]            # corner rows (16 logits each) per LOD
    D, F = 16, 5

    def bytes_fn(S, R):
        """Algorithmic bytes per launch.  A corner row is shared by every sample of the up to eight cells around it (16 samples
        per cell), so the rows are charged ONCE per launch - at most min(8 S, rows of the LOD) of them - not once per sample.
          decode_rows    (one launch per LOD) reads the logits of every row, writes the decoded 5-vector
          trilinear fwd  (one launch, all LODs) per sample: coordinates 12 + per LOD (voxel 8 + 8 trinkets x 4 + 8 decoded rows x 20
                         gathered) + output 20
          bwd            (one entry point, all LODs: magnitude pass + scatter + row pass) per sample: coordinates 12 + output gradient
                         20 + per LOD (voxel 8 + trinkets 32); per touched row: logits read 64 + logit gradients read and written 128;
                         dictionary gradients L x 16 x 5 x 4 (negligible)
        The backward is NOT an HBM-bound kernel: its tables (0.19 M rows) sit in L2 / Infinity Cache and its time is the wave-level run
        merge (VALU) plus ~2.5 M row-wide 64-bit atomic requests; the fraction below is reported against HBM only because SURVEY 8(d)
        prescribes the algorithmic-byte model."""
        touched = [min(8 * S, r) for r in rows]
        return {"codebook_decode_rows": sum(r * (D * 4 + F * 4) for r in rows) / L,
                "spc_trilinear_multi_fwd": (12 + L * (8 + 8 * 4 + 8 * F * 4) + F * 4) * S,
                "codebook_trilinear_multi_bwd": (12 + F * 4 + L * (8 + 8 * 4)) * S + sum(t * 3 * D * 4 for t in touched) + L * D * F * 4,
                "raymarch_voxel_emit": 37 * S, "composite_fwd": 25 * S, "composite_bwd": 41 * S}
    return _nerf_run(args, dev, pipe, tr, bank, args.steps, args.warmup,
                     f"C5: VQAD CodebookOctreeGrid F=5, {L} LODs (levels 5-8), 4-bit codebooks over from_pointcloud(level 8) "
                     f"({int(blas.pyramid[0, 8])} cells), decoders hidden 64 without bias, 'voxel' 16, white background, RMSprop, L2",
                     "training rays/sec, VQAD CodebookOctreeGrid radiance field (LOD 7 = finest of 4), synthetic V8", bytes_fn)


def run_nglod(args, dev):
    """C3: SDF regression steps (512 coordinates each) + sphere-traced rendering of the trained field."""
    import synlego
    import wisp._C as C
    from wisp.core import Rays
    from wisp.tracers import PackedSDFTracer
    from wisp.trainers import SDFTrainStep
    _, blas, grid, nef = nglod_scene(dev)
    tr = SDFTrainStep(nef, lr=1e-3, eps=1e-15, weight_decay=0.0, grid_lr_weight=1.0)
    coords, gts = synlego.armadillo_training_samples(500000, device=dev)
    # keep coordinates inside occupied cells (OctreeSampledSDFDataset samples the octree: every point has features)
    inside = blas.query(coords, 7).pidx >= 0
    coords, gts = coords[inside].contiguous(), gts[inside].contiguous()
    B = args.sdf_batch
    gen = torch.Generator(device=dev).manual_seed(7)

    def batch():
        idx = torch.randint(0, coords.shape[0], (B,), device=dev, generator=gen)
        return C.gather_rows(idx, [coords, gts])

    for _ in range(args.pretrain + args.warmup):
        x, y = batch()
        tr.step(x, y)
    # eager issue (Python + autograd per launch), with the per-launch HIP-event table
    C.TIMING_ALL = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, y = batch()
        loss = tr.step(x, y)
    torch.cuda.synchronize()
    eager_elapsed = time.perf_counter() - t0
    sink, C.TIMING_ALL = C.TIMING_ALL, None
    kernels = _kernel_table(sink)
    # the reference's own trainer class over this package (app/nglod unchanged): SDFTrainer.iterate(), torch.optim.Adam, fp16 autocast
    import copy
    from wisp.datasets import SDFTensorDataset
    from wisp.models import Pipeline
    from wisp.trainers import SDFTrainer, ConfigSDFTrainer, ConfigAdam, ConfigDataloader
    twin = copy.deepcopy(nef)
    dcfg = ConfigSDFTrainer(optimizer=ConfigAdam(lr=1e-3, eps=1e-15), dataloader=ConfigDataloader(batch_size=B), grid_lr_weight=1.0,
                            max_epochs=10 ** 6, enable_amp=True, only_last=True)
    dtr = SDFTrainer(dcfg, Pipeline(twin, None), SDFTensorDataset(coords, gts), device=dev)
    dtr.is_optimization_running = True
    for _ in range(20):
        dtr.iterate()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dsteps = min(args.steps, 200)
    for _ in range(dsteps):
        dtr.iterate()
    torch.cuda.synchronize()
    dropin_elapsed = time.perf_counter() - t0
    del dtr, twin
    # the same steps replayed from a captured HIP graph (fixed batch size: every shape of the step is static)
    tr.capture(B)
    static = tr.static_inputs()

    def batch_into_graph():                                  # the loader gathers straight into the captured graph's input buffers
        idx = torch.randint(0, coords.shape[0], (B,), device=dev, generator=gen)
        return C.gather_rows(idx, [coords, gts], out=list(static))

    for _ in range(args.warmup):
        x, y = batch_into_graph()
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, y = batch_into_graph()
        loss = tr.step(x, y)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # Algorithmic bytes of the step's launches (B = 512 coordinates; 6 LODs of 16 fp32 features; decoder 19 -> 128 -> 1):
    # lookups gather 8 corner rows of 64 B per LOD, the backward writes them back as gradients; the decoder reads its 2.7 K
    # weights once per launch.  With 512 coordinates a launch moves ~1.7 MB: the step is bound by launch latency, which is what
    # the small fractions below say (and why the step is replayed from a captured graph).
    LODS, FW = 6, 16
    nglod_bytes = {"spc_trilinear_multi_fwd": (12 + 8 * LODS + LODS * 8 * (4 + FW * 4) + FW * 4) * B,
                   "spc_trilinear_multi_bwd": (12 + 8 * LODS + LODS * 8 * (4 + 2 * FW * 4) + FW * 4) * B,
                   "small_decoder_fwd": (19 * 4 + 4) * B + (19 * 128 + 128 + 128 + 1) * 4,
                   "small_decoder_bwd": (19 * 4 + 4 + 19 * 4) * B + 2 * (19 * 128 + 128 + 128 + 1) * 4}
    # rendering: 800x800-style rays around the body, 32 marching steps x 0.8 (nglod_octree.yaml tracer)
    o, d, _ = synlego.ray_bank(1 << 18, seed=5, device=dev, with_gt=False)
    rays = Rays(o, d, dist_min=0.0, dist_max=6.0)
    tracer = PackedSDFTracer(num_steps=32, step_size=0.8, min_dis=0.0003)
    rb = tracer(nef, rays=rays, channels=["depth", "hit"], lod_idx=None)
    C.TIMING_ALL = {}
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        rb = tracer(nef, rays=rays, channels=["depth", "hit"], lod_idx=None)
    torch.cuda.synchronize()
    render_s = (time.perf_counter() - t1) / reps
    rsink, C.TIMING_ALL = C.TIMING_ALL, None
    with torch.no_grad():
        err = float((nef(coords=coords[:65536], channels="sdf") - gts[:65536]).abs().mean())
    return {"metric": "SDF training coordinates/sec, NGLOD OctreeGrid (nglod_octree.yaml), synthetic Armadillo", "value": B * args.steps / elapsed,
            "unit": "coords/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3: OctreeGrid F=16, 6 LODs (levels 2-7, 'sum') over from_pointcloud(level 7) of SynArmadillo "
                                   f"({int(blas.pyramid[0, 7])} cells), NeuralSDF 19->128->1, Adam lr 1e-3, batch {B}",
                       "batch": B, "pretrain_steps": args.pretrain},
            "issue": "captured HIP graph (forward + loss + backward) + one optimizer launch per step",
            "eager": {"value": B * args.steps / eager_elapsed, "ms_per_step": 1e3 * eager_elapsed / args.steps},
            "dropin_regime": {"value": B * dsteps / dropin_elapsed, "ms_per_step": 1e3 * dropin_elapsed / dsteps,
                              "note": "wisp.trainers.SDFTrainer.iterate(): the reference trainer's own step (sdf_trainer.py:65-124) - fp16 "
                                      "autocast, autograd, torch.optim.Adam, three .item() read-backs per step"},
            "mean_abs_sdf_error": err, "final_loss": float(loss),
            "render": {"rays": int(o.shape[0]), "ms": 1e3 * render_s, "rays_per_sec": o.shape[0] / render_s,
                       "hit_fraction": float(rb.hit.float().mean()), "marching_steps": 32,
                       "kernels": _kernel_table(rsink)},
            "gpu_busy_fraction_eager": min(1.0, _busy_union_ms(sink)[0] * 1e-3 / eager_elapsed),
            "roofline": _latency_roofline(kernels, nglod_bytes, args.steps, 1e3 * elapsed / args.steps), "kernels": kernels}


def secondary_lines(args, dev, budget_s=6.0):
    """Short in-process runs of the other BASELINE.json configurations for bench.py's default line (`configs`): same code as
    `--config ...`, a handful of steps each, only the summary fields.  The full lines (kernel tables) come from `--config`.
    The radiance-field configs run at the reference trainer's batch (`--target-samples`, 2^18 packed samples per step: `value`) and
    again at `--large-target-samples` (`large_batch_regime`: the figure these lines carried until round 5)."""
    import types
    out = {}

    def summary(r, t0):
        top = list(r["kernels"].items())[:4]
        line = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "roofline") if k in r}
        line["workload"] = r["config"]["workload"]
        for k in ("gpu_busy_fraction", "gpu_busy_fraction_eager", "samples_per_sec", "psnr_db_train_rays", "eager", "issue", "dropin_regime"):
            if k in r:
                line[k] = r[k]
        if "samples_per_step" in r["config"]:
            line["samples_per_step"] = r["config"]["samples_per_step"]
        if "render" in r:
            line["render"] = {k: r["render"][k] for k in ("rays", "ms", "rays_per_sec", "marching_steps")}
        line["top_launches"] = {k: {"avg_ms": v["avg_ms"], "share": v["share"], "launches": v["launches"]} for k, v in top}
        line["wall_s"] = time.perf_counter() - t0
        return line

    for name, fn, over in (("v8", run_v8, dict(pretrain=30, steps=30, warmup=3)),
                           ("vqad", run_vqad, dict(pretrain=30, steps=30, warmup=3)),
                           ("nglod", run_nglod, dict(pretrain=40, steps=200, warmup=5))):
        a = types.SimpleNamespace(**vars(args))
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        try:
            out[name] = summary(fn(a, dev), t0)
            large = int(getattr(args, "large_target_samples", 0) or 0)
            if name != "nglod" and large > 0 and large != a.target_samples:
                a.target_samples = large
                t1 = time.perf_counter()
                big = summary(fn(a, dev), t1)
                out[name]["large_batch_regime"] = {k: big[k] for k in ("value", "unit", "ms_per_step", "samples_per_step", "roofline",
                                                                        "gpu_busy_fraction", "top_launches", "wall_s") if k in big}
        except Exception as e:                                  # a secondary line must not take the headline down
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return out


def main(args, dev, emit=print):
    out = {"v8": run_v8, "vqad": run_vqad, "nglod": run_nglod}[args.config](args, dev)
    emit(json.dumps(out))
    return out
