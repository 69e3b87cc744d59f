"""SDFTrainStep: the optimisation step of the reference's SDFTrainer (wisp/trainers/sdf_trainer.py:65-124) without its app
plumbing - regression of a signed-distance field on (coordinate, distance) pairs:

    loss = sum over the loss LODs of  sum((nef(coords, lod)['sdf'] - gts)^2) / batch      (only_last: the finest LOD alone)

Parameter groups, learning-rate weighting and the fused single-launch optimizer over one flat parameter buffer are the
MultiviewTrainStep's (base_trainer.py:205-235); nglod_octree.yaml trains with Adam, lr 1e-3, eps 1e-15, grid lr x 1."""
import logging as log
from dataclasses import dataclass

import torch

from wisp.trainers.base_trainer import BaseTrainer, ConfigBaseTrainer
from wisp.trainers.multiview_trainer import FlatParams


def _hip():
    import wisp._C as _C
    return _C


class SDFTrainStep:
    def __init__(self, nef, lr=1e-3, eps=1e-15, weight_decay=0.0, grid_lr_weight=1.0, betas=(0.9, 0.999), optimizer='adam',
                 only_last=True, alpha=0.99, momentum=0.0):
        self.nef = nef
        self.flat = FlatParams(nef)
        self.lr, self.eps, self.weight_decay, self.grid_lr_weight, self.betas = lr, eps, weight_decay, grid_lr_weight, betas
        self.optimizer = str(optimizer).lower()
        if self.optimizer not in ('adamw', 'adam', 'rmsprop'):
            raise ValueError(f"optimizer must be 'adamw', 'adam' or 'rmsprop', got {optimizer!r}")
        self.alpha, self.momentum = alpha, momentum
        self.only_last = only_last
        self.opt_steps = 0

    def loss_lods(self):
        lods = list(range(self.nef.grid.num_lods))
        return lods[-1:] if self.only_last else lods

    def optimizer_step(self):
        C = _hip()
        self.opt_steps += 1
        f = self.flat
        groups = []
        for g, lr in (("decoder", self.lr), ("grid", self.lr * self.grid_lr_weight), ("rest", self.lr)):
            a, b = f.ranges[g]
            if b > a:
                groups.append((a, b - a, lr, self.weight_decay, None))
        if self.optimizer == 'rmsprop':
            C.optim_step_groups('rmsprop', f.data, f.grad, f.exp_avg if self.momentum > 0 else None, f.exp_avg_sq, groups,
                                self.alpha, self.momentum, self.eps, self.opt_steps, zero_grad=True)
        else:
            C.optim_step_groups(self.optimizer, f.data, f.grad, f.exp_avg, f.exp_avg_sq, groups, self.betas[0], self.betas[1],
                                self.eps, self.opt_steps, zero_grad=True)

    def _fused_field(self):
        """What wisp_sdf_train_step needs, or None when this field is not its shape: NeuralSDF's decoder (one hidden relu layer,
        biases, one output) over [position, 'sum' OctreeGrid features of 16 channels], fp32 parameters living in the flat
        buffers, loss on the finest LOD only.  WISP_SDF_TRAIN_FUSED=0 keeps the modular launches."""
        import os
        if getattr(self, "_fused_seen_only_last", None) != self.only_last:        # toggled since the decision was taken
            self._fused_cache, self._fused_seen_only_last = None, self.only_last
        cached = getattr(self, "_fused_cache", None)
        if cached is not None:
            # the expensive shape checks are cached; what can change under a live trainer is re-checked on every call: the loss
            # selection, the field's grid / decoder objects, the number of levels, gradients set to None or re-homed
            if cached and not self._fused_still_valid(cached):
                cached = self._fused_cache = None                 # re-derive below
            else:
                return cached or None
        self._fused_cache = False
        if os.environ.get("WISP_SDF_TRAIN_FUSED", "1") == "0" or not self.only_last:
            return None
        from wisp.accelstructs import OctreeAS
        from wisp.models.grids import OctreeGrid
        from wisp.models.nefs._grid_mlp import _fusable_small_decoder
        nef = self.nef
        grid, dec = getattr(nef, "grid", None), getattr(nef, "decoder", None)
        if type(grid) is not OctreeGrid or type(getattr(grid, "blas", None)) is not OctreeAS or dec is None:
            return None
        if not (grid.interpolation_type == 'linear' and grid.multiscale_type == 'sum' and grid.feature_dim == 16 and grid._fusable()
                and 1 <= grid.num_lods <= 16 and type(getattr(nef, "pos_embedder", None)) is torch.nn.Identity
                and getattr(nef, "position_input", False)):
            return None
        probe = torch.empty(1, 3 + grid.feature_dim, device=self.flat.data.device)
        if not probe.is_cuda or not _fusable_small_decoder(dec, probe) or dec.layers[0].in_features != 3 + grid.feature_dim:
            return None
        prm = list(grid.features[:grid.num_lods]) + [dec.layers[0].weight, dec.layers[0].bias, dec.lout.weight, dec.lout.bias]
        if not all(q.is_cuda and q.dtype == torch.float32 and q.is_contiguous() and q.grad is not None and q.grad.is_contiguous()
                   and q.grad.dtype == torch.float32 for q in prm):
            return None
        self._fused_cache = dict(grid=grid, dec=dec, lods=grid.num_lods, prm=prm)
        return self._fused_cache

    def _fused_still_valid(self, c):
        nef = self.nef
        grid, dec = c["grid"], c["dec"]
        if not self.only_last or getattr(nef, "grid", None) is not grid or getattr(nef, "decoder", None) is not dec \
                or grid.num_lods != c["lods"]:
            return False
        now = list(grid.features[:grid.num_lods]) + [dec.layers[0].weight, dec.layers[0].bias, dec.lout.weight, dec.lout.bias]
        if len(now) != len(c["prm"]):
            return False
        for q, was in zip(now, c["prm"]):
            g = q.grad
            if q is not was or g is None or g.dtype != torch.float32 or not g.is_contiguous() or not q.is_contiguous():
                return False
        return True

    def _forward_backward(self, coords, gts):
        fused = self._fused_field() if coords.is_cuda and coords.ndim == 2 and coords.shape[0] > 0 else None
        if fused is not None:
            C = _hip()
            grid, dec = fused["grid"], fused["dec"]
            L = grid.num_lods
            blas = grid.blas
            grid._sync_device(coords.device)
            tr = grid.trinkets if grid.trinkets.dtype == torch.int32 else grid.trinkets.int()
            w1, b1, w2, b2 = dec.layers[0].weight, dec.layers[0].bias, dec.lout.weight, dec.lout.bias
            loss = C.sdf_train_step(coords, gts, blas.octree, blas.prefix, blas.points, tr, [f.detach() for f in grid.features[:L]],
                                    grid.active_lods[:L], grid.half_features, w1.detach(), b1.detach(), w2.detach(), b2.detach(),
                                    [f.grad for f in grid.features[:L]], w1.grad, b1.grad, w2.grad, b2.grad)
            return loss[0]
        loss = 0.0
        for lod_idx in self.loss_lods():
            pred = self.nef(coords=coords, lod_idx=lod_idx, channels="sdf")
            loss = loss + ((pred - gts) ** 2).sum()
        loss = loss / coords.shape[0]
        loss.backward()
        return loss.detach()

    def capture(self, batch_size):
        """Capture forward + loss + backward for a FIXED batch size as one HIP graph (torch.cuda.CUDAGraph on ROCm).
        The step of nglod_octree.yaml is 512 coordinates: a dozen small launches whose GPU time (~50 us) is a fraction of
        what Python + autograd need to issue them (~0.8 ms).  Every shape in it is static - query, trilinear blend and
        decoder see [batch_size, ...] tensors, nothing depends on data - so the launches are recorded once and replayed;
        the fused optimizer stays outside (its bias correction changes every step and it is one launch anyway).
        step() uses the graph whenever it is handed a batch of the captured size; results equal the eager step's."""
        dev = self.flat.data.device
        if dev.type != 'cuda':
            raise RuntimeError("SDFTrainStep.capture needs the model on the GPU")
        self._fused_cache = None                               # the graph bakes pointers in: decide afresh what gets captured
        self._g_coords = torch.zeros(batch_size, 3, dtype=torch.float32, device=dev)
        self._g_gts = torch.zeros(batch_size, 1, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                          # warm-up off the default stream (allocator, lazy set-up)
            for _ in range(3):
                self._forward_backward(self._g_coords, self._g_gts)
        torch.cuda.current_stream().wait_stream(side)
        self.flat.grad.zero_()                                 # the warm-up passes accumulated gradients; parameters untouched
        graph = torch.cuda.CUDAGraph()
        # (captured on the warm-up stream: the per-(device, stream) scratch the ops keep was created - and zeroed - there; on another
        #  stream they would allocate it anew INSIDE the capture, and every replay would zero a few MB again)
        with torch.cuda.graph(graph, stream=side):
            self._g_loss = self._forward_backward(self._g_coords, self._g_gts)
        self.flat.grad.zero_()
        self._graph = graph
        return self

    def static_inputs(self):
        """(coords [B,3], gts [B,1]) buffers of the captured graph, or None: a loader that fills THESE and hands them to step()
        spares the two device copies step() otherwise makes into them."""
        if getattr(self, "_graph", None) is None:
            return None
        return self._g_coords, self._g_gts

    def step(self, coords, gts):
        """coords [B,3], gts [B,1] on the GPU -> loss tensor (already divided by the batch size, like the reference)."""
        graph = getattr(self, "_graph", None)
        if graph is not None and tuple(coords.shape) == tuple(self._g_coords.shape) and tuple(gts.shape) == tuple(self._g_gts.shape):
            if coords is not self._g_coords:
                self._g_coords.copy_(coords)
            if gts is not self._g_gts:
                self._g_gts.copy_(gts)
            graph.replay()
            self.optimizer_step()
            return self._g_loss.clone()
        loss = self._forward_backward(coords, gts)
        self.optimizer_step()
        return loss


@dataclass
class ConfigSDFTrainer(ConfigBaseTrainer):
    """Field names and defaults of wisp/trainers/sdf_trainer.py:20-29 (the YAML schema under `trainer:` of app/nglod)."""
    log_2d: bool = False
    only_last: bool = True
    resample: bool = False


class SDFTrainer(BaseTrainer):
    """The reference's SDF trainer as app/nglod uses it (wisp/trainers/sdf_trainer.py:32-135): the unchanged-trainer regime for
    signed-distance fields - BaseTrainer's life cycle (autocast around step() when enable_amp), autograd over the field, a
    torch.optim optimizer with the name-matched parameter groups.  Events of step() in the reference's order: zero the
    gradients, one field query per loss LOD, sum of squared errors (plus the colour term when the dataset samples textures),
    three metric read-backs, division by the batch size, backward, optimizer step - no GradScaler here (the reference's SDF step
    never scales, sdf_trainer.py:120-123).  SDFTrainStep is the fused MI355X step with the same arithmetic."""

    def __init__(self, cfg, pipeline, train_dataset, tracker=None, device='cuda', scene_state=None):
        super().__init__(cfg=cfg, pipeline=pipeline, train_dataset=train_dataset, tracker=tracker, device=device,
                         scene_state=scene_state)

    def pre_training(self):
        super().pre_training()
        self.tracker.metrics.define_metric('rgb_loss', aggregation_type=float)
        self.tracker.metrics.define_metric('l2_loss', aggregation_type=float)

    def pre_epoch(self):
        super().pre_epoch()
        lods = list(range(self.pipeline.nef.grid.num_lods))
        self.loss_lods = lods[-1:] if self.cfg.only_last else lods

    def post_epoch(self):
        super().post_epoch()
        if self.cfg.resample:
            self.resample_dataset()

    def step(self, data):
        pts = data['coords'].to(self.device)
        gts = data['sdf'].to(self.device)
        with_colour = bool(getattr(self.train_dataset, 'sample_tex', False))
        rgb = data['rgb'].to(self.device) if with_colour else None
        batch_size = pts.shape[0]
        self.pipeline.zero_grad()
        nef = self.pipeline.nef
        loss, l2_loss, rgb_loss = 0, 0.0, 0.0
        last_l2, last_rgb = 0.0, None
        if with_colour:
            preds = [nef(coords=pts, lod_idx=lod_idx, channels=["rgb", "sdf"]) for lod_idx in self.loss_lods]
            for colour, dist in preds:
                last_rgb = ((colour - rgb[..., :3]) ** 2).sum()
                rgb_loss += last_rgb
                last_l2 = ((dist - 1.0 * gts) ** 2).sum()
                l2_loss += last_l2
                loss += rgb_loss                                    # (accumulated inside the loop, as the reference does)
        else:
            preds = [nef(coords=pts, lod_idx=lod_idx, channels=["sdf"])[0] for lod_idx in self.loss_lods]
            for dist in preds:
                last_l2 = ((dist - 1.0 * gts) ** 2).sum()
                l2_loss += last_l2
        loss += l2_loss
        m = self.tracker.metrics
        m.total_loss += loss.item()
        m.l2_loss += last_l2.item()
        if last_rgb is not None:
            m.rgb_loss += last_rgb.item()
        m.num_samples += batch_size
        loss /= batch_size
        loss.backward()
        self.optimizer.step()

    def validate(self):
        return None

    def log_console(self):
        m = self.tracker.metrics
        log.info('EPOCH {}/{} | total loss: {:>.3E} | l2 loss: {:>.3E} | rgb loss: {:>.3E}'.format(
            self.epoch, self.max_epochs, m.average_metric('total_loss'), m.average_metric('l2_loss'), m.average_metric('rgb_loss')))
