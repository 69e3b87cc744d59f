"""Oracle (test infrastructure): the radiance-field decoder, the packed tracer and one training step,
torch-CPU.

Restates wisp/models/nefs/nerf.py:151-277 (decoders, rgba, prune), wisp/models/decoders/basic_decoders.py:59-101,
wisp/models/embedders/positional_embedder.py:18-66, wisp/tracers/packed_rf_tracer.py:107-181 and the caller
semantics of wisp/trainers/multiview_trainer.py:111-180 / wisp/trainers/base_trainer.py:205-246.
Parameter names equal the reference's state-dict names so weights can be copied between this oracle and
the HIP-backed classes with load_state_dict.

Parity: PINNED to the reference's own code executed on the host - embedder / decoder modules, the rgba and prune method bodies, the
trainer bodies, PackedRFTracer.trace, and end to end: render and three AdamW steps through the reference's OctreeAS -> HashGrid (its
kernel bodies) -> NeuralRadianceField -> PackedRFTracer (tests/test_reference_modules.py) - over the unpinned Kaolin leaves.
The fp16-autocast emulation (rgba(autocast_half=True), train_step(scaler=...)) is pinned the same way: the same reference stack
executed under torch.autocast('cpu', dtype=torch.float16) with HashGridInterpolate's `codebook.half()` branch taken - per-sample
colours / densities bit for bit, loss, decoder gradients to one fp16 ulp (test_half_rounding_oracle_equals_the_reference_stack_
under_fp16_autocast_on_the_host).  Deliberately different: the table gradient is accumulated in fp32 (the reference rounds every
__half2 atomicAdd in arrival order, which nobody can restate).
"""
import numpy as np
import torch
import torch.nn as nn

from . import spc, raymarch, hashgrid, render


class OracleDecoder(nn.Module):
    """BasicDecoder (basic_decoders.py:59-101) with relu + nn.Linear, no skips."""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, bias):
        super().__init__()
        dims = [input_dim] + [hidden_dim] * num_layers
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=bias) for i in range(num_layers)])
        self.lout = nn.Linear(hidden_dim, output_dim, bias=bias)

    def forward(self, x):
        h = x
        for l in self.layers:
            h = torch.relu(l(h))
        return self.lout(h)


def positional_embed(x, num_freq, include_input=True):
    """PositionalEmbedder.forward (positional_embedder.py:51-66): bands 2^0..2^(F-1);
    layout [x ; sin(band_k * x) frequency-major axis-minor ; cos(same)]."""
    bands = 2.0 ** torch.linspace(0.0, num_freq - 1, steps=num_freq, dtype=x.dtype)
    winded = (x[:, None] * bands[None, :, None]).reshape(x.shape[0], x.shape[1] * num_freq)
    enc = torch.cat([torch.sin(winded), torch.cos(winded)], dim=-1)
    return torch.cat([x, enc], dim=-1) if include_input else enc


class _Table(nn.Module):
    def __init__(self, total, F, std):
        super().__init__()
        self.feats = nn.Parameter(torch.randn(total, F) * std)


class _Grid(nn.Module):
    def __init__(self, total, F, std):
        super().__init__()
        self.codebook = _Table(total, F, std)


class OracleNeRF(nn.Module):
    """NeuralRadianceField over a HashGrid (nerf.py:30-277) - the nerf_hash.yaml shape by default."""

    def __init__(self, resolutions, feature_dim=2, codebook_bitwidth=19, multiscale_type='cat', feature_std=1e-9,
                 hidden_dim=64, num_layers=1, bias=True, view_multires=4, coord_dim=3):
        super().__init__()
        self.resolutions = [int(r) for r in resolutions]
        self.feature_dim, self.bitwidth, self.multiscale_type = feature_dim, codebook_bitwidth, multiscale_type
        self.view_multires = view_multires
        sizes, begin = hashgrid.table_layout(self.resolutions, 2 ** codebook_bitwidth, coord_dim)
        self.begin_idxes = torch.from_numpy(begin)
        self.grid = _Grid(int(begin[-1]), feature_dim, feature_std)
        L = len(self.resolutions)
        eff = feature_dim * L if multiscale_type == 'cat' else feature_dim
        self.decoder_density = OracleDecoder(eff, 16, hidden_dim, num_layers, bias)
        if bias:
            self.decoder_density.lout.bias.data[0] = 1.0                      # nerf.py:162-163
        self.view_embed_dim = 3 + 3 * 2 * view_multires
        self.decoder_color = OracleDecoder(15 + self.view_embed_dim, 3, hidden_dim, num_layers + 1, bias)

    def rgba(self, coords, ray_d, lod_idx=None, autocast_half=False):
        """nerf.py:219-264.  autocast_half: round where the reference rounds when BaseTrainer.iterate wraps the step in
        `torch.cuda.amp.autocast()` (base_trainer.py:338, fp16): the table is cast to half and the lookup returns half
        (ops/grid.py:77-89: custom_fwd + `codebook.half()`), every nn.Linear runs on half operands and returns half (torch's
        autocast policy; relu / sigmoid follow their input), the view embedding stays fp32 and `cat` promotes.  The table
        gradient is accumulated in fp32 from the half upstream gradient (the reference's __half2 atomics round every add in
        an order nobody can restate)."""
        L = len(self.resolutions)
        if lod_idx is None:
            lod_idx = L - 1
        table = self.grid.codebook.feats
        if autocast_half:
            table = table + (table.half().float() - table).detach()        # value of codebook.half(), gradient of the cast
        feats = hashgrid.grid_interpolate(coords, lod_idx, self.multiscale_type, self.feature_dim, self.resolutions,
                                          self.bitwidth, table, self.begin_idxes)
        if not autocast_half:
            feats = feats.to(self.decoder_density.lout.weight.dtype)
            density_feats = self.decoder_density(feats)
            emb = positional_embed(ray_d.to(feats.dtype), self.view_multires, include_input=True)
            fdir = torch.cat([density_feats, emb], dim=-1)
            colors = torch.sigmoid(self.decoder_color(fdir[..., 1:]))
            density = torch.relu(density_feats[..., 0:1])
            return dict(rgb=colors, density=density)
        feats = feats.half()                                                  # the kernel's output dtype is the table's
        with torch.autocast('cpu', dtype=torch.float16):
            density_feats = self.decoder_density(feats)
            emb = positional_embed(ray_d.float(), self.view_multires, include_input=True)
            fdir = torch.cat([density_feats, emb], dim=-1)
            colors = torch.sigmoid(self.decoder_color(fdir[..., 1:]))
            density = torch.relu(density_feats[..., 0:1])
        return dict(rgb=colors.float(), density=density.float())           # compositing runs in fp32 on the half values


class OracleBLAS:
    """The octree acceleration structure state (octree_as.py:42-63) as numpy arrays."""

    def __init__(self, octree):
        self.octree = np.asarray(octree, dtype=np.uint8)
        self.points, self.pyramid, self.exsum = spc.octree_to_spc(self.octree)
        self.max_level = self.pyramid.shape[1] - 2

    @classmethod
    def make_dense(cls, level):
        return cls(spc.create_dense_octree(level))

    @classmethod
    def from_quantized_points(cls, pts, level):
        return cls(spc.points_to_octree(pts, level))

    def level_points(self):
        s, n = int(self.pyramid[1, self.max_level]), int(self.pyramid[0, self.max_level])
        return self.points[s:s + n]


def trace(nef, blas, origins, dirs, near, far, num_steps, jitter, bg_color=(0.0, 0.0, 0.0), raymarch_type='ray',
          with_depth=True, lod_idx=None, autocast_half=False):
    """PackedRFTracer.trace (packed_rf_tracer.py:107-181) for channels rgb/alpha/depth/hit."""
    o = origins.detach().cpu().numpy()
    d = dirs.detach().cpu().numpy()
    if raymarch_type == 'ray':
        rm = raymarch.raymarch_ray(blas.octree, blas.exsum, o, d, near, far, num_steps, blas.max_level, jitter)
    elif raymarch_type == 'voxel':
        rm = raymarch.raymarch_voxel(blas.octree, blas.points, blas.pyramid, blas.exsum, o, d, num_steps,
                                     blas.max_level, jitter)
    elif raymarch_type == 'uniform':
        rm = raymarch.raymarch_uniform(blas.octree, blas.points, blas.pyramid, blas.exsum, o, d, num_steps,
                                       blas.max_level)
    else:
        raise TypeError(f"Raymarch sampler type: {raymarch_type} is not supported by OctreeAS.")
    ridx = torch.from_numpy(rm["ridx"])
    samples = torch.from_numpy(rm["samples"])
    deltas = torch.from_numpy(rm["deltas"])
    depths = torch.from_numpy(rm["depth_samples"])
    boundary = torch.from_numpy(rm["boundary"])
    hit_ray_d = dirs.index_select(0, ridx)
    out = nef.rgba(samples, hit_ray_d, lod_idx, autocast_half) if autocast_half else nef.rgba(samples, hit_ray_d, lod_idx)
    res = render.composite(out["rgb"], out["density"], deltas, depths, ridx, boundary, origins.shape[0], bg_color,
                           with_depth=with_depth)
    res["raymarch"] = rm
    res["sample_rgb"] = out["rgb"]
    res["sample_density"] = out["density"]
    return res


def psnr(rgb, gts):
    """wisp/ops/image/metrics.py:19-37."""
    mse = torch.mean((rgb[..., :3] - gts[..., :3]) ** 2).item()
    return 10 * np.log10(1.0 / mse)


def make_optimizer(nef, lr=1e-3, eps=1e-16, weight_decay=1e-6, grid_lr_weight=500.0):
    """BaseTrainer.init_optimizer (base_trainer.py:205-235): 'decoder' names -> weight decay group,
    'grid' names -> lr * grid_lr_weight; AdamW(lr, eps, weight_decay) defaults fill the rest."""
    dec, grid, rest = [], [], []
    for name, p in nef.named_parameters():
        (dec if 'decoder' in name else grid if 'grid' in name else rest).append(p)
    groups = [{"params": dec, "lr": lr, "eps": eps, "weight_decay": weight_decay},
              {"params": grid, "eps": eps, "lr": lr * grid_lr_weight},
              {"params": rest, "eps": eps, "lr": lr}]
    return torch.optim.AdamW(groups, lr=lr, eps=eps, weight_decay=weight_decay)


def make_scaler(init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
    """The state of torch.cuda.amp.GradScaler at its defaults (base_trainer.py:190: `GradScaler()`)."""
    return dict(scale=float(init_scale), growth=float(growth_factor), backoff=float(backoff_factor), interval=int(growth_interval),
                good_steps=0, skipped=0)


def train_step(nef, blas, optimizer, origins, dirs, gts, near, far, num_steps, jitter, bg_color=(0.0, 0.0, 0.0),
               raymarch_type='ray', loss_type='huber', scaler=None):
    """MultiviewTrainer.step (multiview_trainer.py:111-180): trace, loss over rays, backward, step.
    scaler (make_scaler()): the enable_amp regime - the forward rounds where fp16 autocast rounds (OracleNeRF.rgba), the loss
    is scaled, the gradients are unscaled in fp32, a non-finite gradient skips the step and halves the scale
    (multiview_trainer.py:168-171 + GradScaler.step / update)."""
    optimizer.zero_grad()
    res = trace(nef, blas, origins, dirs, near, far, num_steps, jitter, bg_color, raymarch_type, with_depth=False,
                autocast_half=scaler is not None)
    if loss_type == 'huber':
        loss = torch.nn.functional.smooth_l1_loss(res["rgb"], gts, reduction='none').mean()
    elif loss_type == 'l2':
        loss = torch.nn.functional.mse_loss(res["rgb"], gts, reduction='none').mean()
    else:
        loss = torch.abs(res["rgb"] - gts).mean()
    if scaler is None:
        loss.backward()
        optimizer.step()
        return float(loss.detach()), int(res["raymarch"]["ridx"].shape[0])
    (loss * scaler["scale"]).backward()
    params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    finite = all(bool(torch.isfinite(p.grad).all()) for p in params)
    if finite:
        inv = 1.0 / scaler["scale"]
        for p in params:
            p.grad.mul_(inv)
        optimizer.step()
        scaler["good_steps"] += 1
        if scaler["good_steps"] == scaler["interval"]:
            scaler["scale"] *= scaler["growth"]
            scaler["good_steps"] = 0
    else:
        scaler["scale"] *= scaler["backoff"]
        scaler["good_steps"] = 0
        scaler["skipped"] += 1
    return float(loss.detach()), int(res["raymarch"]["ridx"].shape[0])


def prune(nef, blas, occupancy, dense_points, density_decay, min_density, unit_samples, view_dirs):
    """NeuralRadianceField.prune (nerf.py:175-212) with the random draws injected:
    unit_samples [P,3] in [0,1), view_dirs [P,3].  Returns (new_blas or None, new occupancy)."""
    occupancy = occupancy * density_decay
    res = 2.0 ** blas.max_level
    pts = torch.from_numpy(dense_points.astype(np.float32))
    samples = ((pts + unit_samples) / res) * 2.0 - 1.0
    with torch.no_grad():
        density = nef.rgba(samples, view_dirs)["density"]
    occupancy = torch.stack([density[:, 0], occupancy], -1).max(dim=-1)[0]
    mask = (occupancy > min_density).numpy()
    kept = dense_points[mask]
    if kept.shape[0] == 0:
        return None, occupancy
    return OracleBLAS.from_quantized_points(kept, blas.max_level), occupancy
