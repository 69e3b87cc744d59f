"""CPU, world_size 2, gloo: the data-parallel logic of the N>1 path (ray sharding, flat-buffer gradient all-reduce with
1/world scaling, identical prune draws on every rank).  The HIP kernels cannot run here, so the per-ray work is stood in
by a small differentiable torch function with the same structure (independent rays, shared parameters, mean-over-rays
loss); what is under test is everything that differs between N=1 and N>1."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class TinyField(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.grid = torch.nn.Embedding(64, 4)                 # name contains 'grid'  -> grid group
        self.decoder_color = torch.nn.Linear(4 + 3, 3)        # name contains 'decoder' -> decoder group
        self.other = torch.nn.Parameter(torch.zeros(3))

    def forward(self, origins, dirs):
        cell = ((origins[:, 0] * 0.5 + 0.5) * 63).long().clamp(0, 63)
        return torch.sigmoid(self.decoder_color(torch.cat([self.grid(cell), dirs], -1))) + self.other


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wisp.trainers import FlatParams, MultiviewTrainStep, shard_rays
    g = torch.Generator().manual_seed(1)
    O, D, T = torch.rand(101, 3, generator=g) * 2 - 1, torch.randn(101, 3, generator=g), torch.rand(101, 3, generator=g)
    model = TinyField()
    flat = FlatParams(model)
    assert list(flat.ranges) == ["decoder", "grid", "rest"]
    assert all(p.data_ptr() >= flat.data.data_ptr() for p in model.parameters())
    # single-process reference gradient over ALL rays
    ref = TinyField()
    torch.nn.functional.smooth_l1_loss(ref(O, D), T, reduction='none').mean().backward()
    ref_flat = torch.cat([p.grad.reshape(-1) for n, p in sorted(ref.named_parameters(), key=lambda kv: (
        0 if 'decoder' in kv[0] else 1 if 'grid' in kv[0] else 2))])
    # this rank's shard; equal shard sizes are required for mean-of-means == global mean, so pad-free split of 100 rays
    lo, hi = shard_rays(100, rank, world)
    torch.nn.functional.smooth_l1_loss(model(O[lo:hi], D[lo:hi]), T[lo:hi], reduction='none').mean().backward()
    step = MultiviewTrainStep.__new__(MultiviewTrainStep)
    step.flat, step.world, step.group, step.force_allreduce = flat, world, None, False
    step.allreduce_grads()
    avg = flat.grad / world                                   # what adamw_step's grad_scale = 1/world applies
    ref100 = TinyField()
    torch.nn.functional.smooth_l1_loss(ref100(O[:100], D[:100]), T[:100], reduction='none').mean().backward()
    ref_flat100 = torch.cat([p.grad.reshape(-1) for n, p in sorted(ref100.named_parameters(), key=lambda kv: (
        0 if 'decoder' in kv[0] else 1 if 'grid' in kv[0] else 2))])
    packed = torch.cat([avg[a:a + p.numel()] for a, p in _segments(flat, model)])
    ok = torch.allclose(packed, ref_flat100, atol=1e-6)
    # prune draws are rank-independent
    gen = torch.Generator().manual_seed(0)
    draw = torch.rand(5, 3, generator=gen)
    gathered = [torch.zeros_like(draw) for _ in range(world)]
    dist.all_gather(gathered, draw)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    shards = [shard_rays(101, r, world) for r in range(world)]
    cover = shards[0][0] == 0 and shards[-1][1] == 101 and all(shards[i][1] == shards[i + 1][0] for i in range(world - 1))
    out[rank] = bool(ok and same and cover)
    dist.barrier()
    dist.destroy_process_group()


def _segments(flat, model):
    """(offset, param) for every trainable parameter in flat-buffer order."""
    base = flat.data.data_ptr()
    return sorted(((p.data_ptr() - base) // 4, p) for p in model.parameters() if p.requires_grad)


def test_ray_sharded_gradient_allreduce_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
