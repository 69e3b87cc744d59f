"""Worker of the data-parallel GPU tests (launched by tests/test_gpu_multi.py under torch.distributed.run, one rank per
GPU, RCCL): real MultiviewTrainStep.step() on the direct-issue flagship path with a prune inside the run.
Prints one line `DP_RESULT {json}` on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]

import torch
import torch.distributed as dist


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    import synlego
    from wisp.accelstructs import OctreeAS
    from wisp.core import Rays
    from wisp.models import Pipeline
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    from wisp.tracers import PackedRFTracer
    from wisp.trainers import MultiviewTrainStep, shard_rays
    level, N, R = 5, 128, 4096
    steps, prune_every = int(os.environ.get("DP_STEPS", "7")), 3
    amp = os.environ.get("DP_AMP", "1") == "1"

    def build():
        torch.manual_seed(0)
        blas = OctreeAS.make_dense(level)
        grid = HashGrid.from_geometric(blas, feature_dim=2, num_lods=16, multiscale_type='cat', feature_std=1e-2,
                                       codebook_bitwidth=14, min_grid_res=8, max_grid_res=128)
        nef = NeuralRadianceField(grid, view_embedder='positional', view_multires=4, hidden_dim=64, num_layers=1, bias=True,
                                  prune_density_decay=0.95, prune_min_density=0.5).to(dev)
        pipe = Pipeline(nef, PackedRFTracer(raymarch_type='ray', num_steps=N, bg_color=(0.0, 0.0, 0.0)))
        return pipe, MultiviewTrainStep(pipe, prune_every=prune_every, enable_amp=amp, lr=1e-2, grid_lr_weight=10.0, seed=3)

    o, d, gt = synlego.ray_bank(R, seed=5, device=dev)          # the same bank on every rank; each takes its shard
    jit = torch.rand(R, N, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    lo, hi = shard_rays(R, rank, world)
    pipe, tr = build()
    used_direct = tr._direct is not None
    for _ in range(steps):
        tr.step(Rays(o[lo:hi], d[lo:hi], dist_min=1.0, dist_max=5.0), gt[lo:hi], jitter=jit[lo:hi])
    tr.wait_for_parameters()
    sharded = bool(tr._plan)
    stale_before_sync = bool(tr._master_stale)
    tr.sync_master()                          # no-op on the all-reduce path; collective on the sharded one (bf16 shadow travelled)
    torch.cuda.synchronize()
    flat = tr.flat.data
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    identical = all(torch.equal(gathered[0], g) for g in gathered)
    oc = pipe.nef.grid.blas.octree
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([oc.numel()], dtype=torch.int64, device=dev))
    same_tree = all(int(s) == int(sizes[0]) for s in sizes)
    if same_tree:
        trees = [torch.empty_like(oc) for _ in range(world)]
        dist.all_gather(trees, oc)
        same_tree = all(torch.equal(trees[0], t) for t in trees)
    pruned = int(pipe.nef.grid.blas.pyramid[0, level]) < 8 ** level
    res = dict(world=world, identical=bool(identical), same_tree=bool(same_tree), pruned=bool(pruned), direct=bool(used_direct),
               finite=bool(torch.isfinite(flat).all()),
               # the all-reduce leaves out the finest level's rows (never a gradient under 'cat' with lod_idx = num_lods - 1):
               # the skipped tail of the gradient buffer must indeed be all zero
               allreduce_numel=int(tr._live_grad_numel()), grad_numel=int(tr.flat.grad.numel()),
               skipped_tail_zero=bool((tr.flat.grad[tr._live_grad_numel():] == 0).all()),
               sharded=sharded, stale_before_sync=stale_before_sync, grads_consumed=bool((tr.flat.grad == 0).all()))
    if sharded:
        ga, gb = tr.flat.ranges["grid"]
        plan = tr._plan
        res.update(plan_direct=bool(plan["direct"]), slice_elems=int(plan["c"]), window=int(plan["npad"]), grid_elems=int(gb - ga))
        if tr.flat.shadow is not None:        # every slice's shadow, and the tail every rank updates itself, mirror the master
            res["shadow_is_bf16_of_master"] = bool(torch.equal(tr.flat.shadow, tr.flat.data[ga:gb].to(torch.bfloat16)))
        own = torch.zeros(tr.flat.data.numel(), dtype=torch.bool, device=dev)
        own[plan["lo"]:plan["hi"]] = True
        window = torch.zeros_like(own)
        window[ga:min(ga + plan["npad"], gb)] = True
        off = tr.flat.exp_avg_sq[window & ~own]
        res["state_only_on_owner"] = bool(off.numel() == 0 or float(off.abs().max()) == 0.0)
    # single-rank reference of the SAME global batch, on rank 0 only, without any collective: the data-parallel result must
    # agree with it up to the summation order of the gradient
    dist.barrier()
    if rank == 0:
        os.environ["WISP_FORCE_ALLREDUCE"] = "0"
        os.environ["WISP_SHARDED_OPTIM"] = "0"
        pipe1, tr1 = build()
        tr1.world, tr1.force_allreduce = 1, False
        for _ in range(steps):
            tr1.step(Rays(o, d, dist_min=1.0, dist_max=5.0), gt, jitter=jit)
        torch.cuda.synchronize()
        ref = tr1.flat.data
        res["max_abs_diff_vs_single"] = float((ref - flat).abs().max())
        res["rel_l2_vs_single"] = float((ref - flat).norm() / ref.norm())
        res["bit_equal_vs_single"] = bool(torch.equal(ref, flat))
        print("DP_RESULT " + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
