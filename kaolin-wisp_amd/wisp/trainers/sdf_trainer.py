"""SDFTrainStep: the optimisation step of the reference's SDFTrainer (wisp/trainers/sdf_trainer.py:65-124) without its app
plumbing - regression of a signed-distance field on (coordinate, distance) pairs:

    loss = sum over the loss LODs of  sum((nef(coords, lod)['sdf'] - gts)^2) / batch      (only_last: the finest LOD alone)

Parameter groups, learning-rate weighting and the fused single-launch optimizer over one flat parameter buffer are the
MultiviewTrainStep's (base_trainer.py:205-235); nglod_octree.yaml trains with Adam, lr 1e-3, eps 1e-15, grid lr x 1."""
import torch

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

    def step(self, coords, gts):
        """coords [B,3], gts [B,1] on the GPU -> loss tensor (already divided by the batch size, like the reference)."""
        loss = 0.0
        for lod_idx in self.loss_lods():
            pred = self.nef(coords=coords, lod_idx=lod_idx, channels="sdf")
            loss = loss + ((pred - gts) ** 2).sum()
        loss = loss / coords.shape[0]
        loss.backward()
        self.optimizer_step()
        return loss.detach()
