"""BaseTrainer: the life cycle an application's trainer subclasses (wisp/trainers/base_trainer.py:93-600) - the part of it
the optimisation step runs through.  This is the *unchanged-application* regime: torch.optim optimizers built by
`init_optimizer` (parameter groups by name, base_trainer.py:205-246), fp16 autocast around `step()` and a GradScaler
(base_trainer.py:240,338), autograd over the modular Pipeline.  The MI355X-specific fused step lives next door
(`MultiviewTrainStep`); a trainer written against the reference's BaseTrainer gets this class.

Out of scope here: scene graph / interactive state, tracker back-ends (tensorboard, wandb), snapshot rendering.
`tracker` may be any object with a `metrics` attribute; `None` gives an in-memory metrics record.
"""
import logging as log
import os
import time
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch


# ---- optimizer / trainer configuration records: field names and defaults of wisp/config/presets/torch.py:45-68 and
# ---- base_trainer.py:22-91 (the YAML schema of app/nerf/configs/*.yaml); `constructor` = the YAML's selector
@dataclass
class ConfigAdam:
    lr: float = 1e-3
    betas: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8
    weight_decay: float = 0.0
    constructor: str = 'Adam'


@dataclass
class ConfigAdamW:
    lr: float = 1e-3
    betas: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8
    weight_decay: float = 0.0
    constructor: str = 'AdamW'


@dataclass
class ConfigRMSprop:
    lr: float = 1e-2
    alpha: float = 0.99
    eps: float = 1e-8
    weight_decay: float = 0.0
    momentum: float = 0.0
    constructor: str = 'RMSprop'


@dataclass
class ConfigDataloader:
    batch_size: int = 1
    num_workers: int = 0


@dataclass
class ConfigBaseTrainer:
    optimizer: object = field(default_factory=ConfigAdamW)
    dataloader: ConfigDataloader = field(default_factory=ConfigDataloader)
    exp_name: str = 'unnamed'
    mode: str = 'train'
    max_epochs: int = 250
    save_every: int = -1
    save_as_new: bool = False
    model_format: str = 'full'
    render_every: int = 100
    valid_every: int = -1
    valid_split: str = 'test'
    enable_amp: bool = True
    profile_nvtx: bool = True
    grid_lr_weight: float = 1.0
    scheduler: bool = False
    scheduler_milestones: Tuple[float, ...] = (0.5, 0.75, 0.9)
    scheduler_gamma: float = 0.333
    valid_metrics: Tuple[str, ...] = ('psnr',)


_OPTIMIZERS = {'adam': torch.optim.Adam, 'adamw': torch.optim.AdamW, 'rmsprop': torch.optim.RMSprop, 'sgd': torch.optim.SGD}


def instantiate_optimizer(cfg, params):
    """`instantiate(cfg.optimizer, params=params)` of the reference's config system (wisp/config/utils.py), reduced to what
    the trainer needs: `cfg.constructor` names a torch.optim class (or is one / any callable), every other public field is a
    keyword argument.  FusedAdam (apex, image_hash.yaml) resolves to torch.optim.Adam - same arithmetic."""
    ctor = getattr(cfg, 'constructor', 'AdamW')
    if isinstance(ctor, str):
        key = ctor.lower()
        key = 'adam' if key == 'fusedadam' else key
        if key not in _OPTIMIZERS:
            raise ValueError(f"unknown optimizer constructor {ctor!r}")
        ctor = _OPTIMIZERS[key]
    fields = vars(cfg) if not isinstance(cfg, dict) else cfg
    kwargs = {k: v for k, v in fields.items() if k != 'constructor' and not k.startswith('_')}
    if ctor in (torch.optim.Adam, torch.optim.AdamW) and 'fused' not in kwargs and 'foreach' not in kwargs \
            and os.environ.get("WISP_TORCH_FUSED_OPTIM", "1") != "0":
        # torch's single-kernel Adam(W) (same update rule; takes the GradScaler's scale / found_inf inside the kernel) instead of
        # its default ~12 multi-tensor passes over the 42 MB table: 0.51 -> 0.06 ms per step at nerf_hash.yaml's size on an MI355X
        tensors = [p for g in params for p in g["params"]]
        if tensors and all(p.is_cuda and p.dtype == torch.float32 for p in tensors):
            kwargs['fused'] = True
    return ctor(params, **kwargs)


class _Metrics:
    """In-memory stand-in for tracker.metrics (wisp/trainers/tracker/metrics.py): named running sums + a sample count."""

    def __init__(self):
        object.__setattr__(self, '_m', {})
        self.clear()

    def define_metric(self, name, aggregation_type=float):
        self._m.setdefault(name, aggregation_type())

    def clear(self):
        for k in list(self._m):
            self._m[k] = type(self._m[k])()
        self._m['total_loss'] = 0.0
        self._m['num_samples'] = 0

    def __getattr__(self, name):
        m = object.__getattribute__(self, '_m')
        if name in m:
            return m[name]
        if name.endswith('loss'):
            return 0.0
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self._m[name] = value

    def average_metric(self, name):
        return self._m.get(name, 0.0) / max(self._m.get('num_samples', 0), 1)


class _Tracker:
    def __init__(self, log_dir=None):
        self.metrics = _Metrics()
        self.log_dir = log_dir or os.path.join('_results', 'logs', 'runs')

    def log_metric(self, *a, **k):
        pass

    def log_artifact(self, *a, **k):
        pass

    def teardown(self):
        pass


class _OneViewLoader:
    """The DataLoader of base_trainer.py:197-203 for datasets that already live in HBM: `batch_size` views per batch in a
    fresh random order every epoch, each passed through the dataset's transform, collated by stacking (a leading batch
    dimension, which MultiviewTrainer.step squeezes again)."""

    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, max(int(batch_size), 1)

    def __len__(self):
        # torch's DataLoader with drop_last=False (base_trainer.py:197-203): the partial last batch counts
        return max(-(-len(self.dataset) // self.batch_size), 1)

    def __iter__(self):
        n = len(self.dataset)
        if self.batch_size > 1 and hasattr(self.dataset, "get_batch"):
            # coordinate datasets (SDF training: 512 points per batch): one indexed read instead of 512 items + a collate
            order = torch.randperm(n, device=getattr(self.dataset, "device", "cpu"))
            for i in range(0, n, self.batch_size):
                yield self.dataset.get_batch(order[i:i + self.batch_size])
            return
        order = torch.randperm(n).tolist()
        for i in range(0, n, self.batch_size):
            items = [self.dataset[j] for j in order[i:i + self.batch_size]]
            yield _collate(items)


def _collate(items):
    from wisp.core import Rays
    first = items[0]
    out = {}
    for k in first:
        v = first[k]
        if v is None:
            continue
        if isinstance(v, Rays):
            out[k] = Rays.stack([it[k] for it in items]) if len(items) > 1 else v[None]
        elif torch.is_tensor(v):
            out[k] = torch.stack([it[k] for it in items]) if len(items) > 1 else v[None]
        else:
            out[k] = [it[k] for it in items]
    return out


class BaseTrainer(ABC):
    """init() -> init_optimizer(), init_dataloader(); train() = while running: iterate(); iterate() = [pre_training]
    [begin_epoch] next_batch, pre_step, autocast(step), post_step, [end_epoch -> post_epoch, validate] [post_training]."""

    def __init__(self, cfg, pipeline, train_dataset, tracker=None, device='cuda', scene_state=None):
        self.device = device
        self.cfg = cfg
        self.pipeline = pipeline.to(device)
        self.train_dataset = train_dataset
        self.tracker = tracker if tracker is not None else _Tracker()
        self.scene_state = scene_state
        self.scaler = None
        self.train_data_loader_iter = None
        self.val_data_loader = None
        self.train_dataset_size = None
        self.enable_amp = cfg.enable_amp
        self.max_epochs = cfg.max_epochs
        self.epoch = 1
        self.iteration = 0
        self.is_optimization_running = False
        self.return_dict = {}
        self.init_optimizer()
        self.init_dataloader()
        log.info(f"Total number of parameters: {sum(p.numel() for p in self.pipeline.nef.parameters())}")

    # ------------------------------------------------------------------------------------------ optimizer / data
    def init_optimizer(self):
        """base_trainer.py:205-246: names containing 'decoder' -> weight decay; else names containing 'grid' -> lr x
        grid_lr_weight (no weight decay entry: the optimizer's default applies); everything else plain."""
        buckets = {'decoder': [], 'grid': [], 'rest': []}
        for name, param in self.pipeline.nef.named_parameters():
            buckets['decoder' if 'decoder' in name else 'grid' if 'grid' in name else 'rest'].append(param)
        oc = self.cfg.optimizer
        params = [dict(params=buckets['decoder'], lr=oc.lr, eps=oc.eps, weight_decay=oc.weight_decay),
                  dict(params=buckets['grid'], eps=oc.eps, lr=oc.lr * self.cfg.grid_lr_weight),
                  dict(params=buckets['rest'], eps=oc.eps, lr=oc.lr)]
        self.optimizer = instantiate_optimizer(oc, params)
        dev_type = torch.device(self.device).type if not isinstance(self.device, torch.device) else self.device.type
        self.scaler = torch.amp.GradScaler(dev_type, enabled=(dev_type == 'cuda' and torch.cuda.is_available()))
        if self.cfg.scheduler:
            max_steps = len(self.train_dataset) * self.cfg.max_epochs
            self.scheduler = torch.optim.lr_scheduler.MultiStepLR(
                self.optimizer, milestones=[max_steps * x for x in self.cfg.scheduler_milestones],
                gamma=self.cfg.scheduler_gamma)

    def init_dataloader(self):
        self.train_data_loader = _OneViewLoader(self.train_dataset, self.cfg.dataloader.batch_size)
        self.iterations_per_epoch = len(self.train_data_loader)

    def reset_data_iterator(self):
        self.train_data_loader_iter = iter(self.train_data_loader)

    def next_batch(self):
        return next(self.train_data_loader_iter)

    def resample_dataset(self):
        if not hasattr(self.train_dataset, 'resample'):
            raise ValueError("resample=True but the training dataset doesn't have a resample method")
        self.train_dataset.resample()
        self.init_dataloader()

    # ------------------------------------------------------------------------------------------ life cycle
    @property
    def total_iterations(self) -> int:
        return (self.epoch - 1) * self.iterations_per_epoch + self.iteration

    @property
    def max_iterations(self) -> int:
        return self.max_epochs * self.iterations_per_epoch

    def is_first_iteration(self):
        return self.total_iterations == 0

    def is_any_iterations_remaining(self):
        return self.total_iterations < self.max_iterations

    def begin_epoch(self):
        self.reset_data_iterator()
        self.pre_epoch()
        self.epoch_start_time = time.time()

    def end_epoch(self):
        self.post_epoch()
        if self.cfg.valid_every > -1 and self.epoch % self.cfg.valid_every == 0 and self.epoch != 0:
            self.validate()
        if self.epoch < self.max_epochs:
            self.iteration = 0
            self.epoch += 1
        else:
            self.is_optimization_running = False

    def iterate(self):
        """One batch (base_trainer.py:316-342).  step() runs under autocast - fp16, torch's default for 'cuda', exactly
        like the reference's `torch.cuda.amp.autocast(self.enable_amp)`."""
        if not self.is_optimization_running:
            return
        if self.is_first_iteration():
            self.pre_training()
        data = None
        try:
            if self.train_data_loader_iter is None:
                self.begin_epoch()
            self.iteration += 1
            data = self.next_batch()
        except StopIteration:
            self.end_epoch()
            if self.is_any_iterations_remaining():
                self.begin_epoch()
                data = self.next_batch()
            else:
                self.post_training()
        if self.is_any_iterations_remaining() and data is not None:
            self.pre_step()
            dev_type = torch.device(self.device).type
            with torch.autocast(dev_type, enabled=bool(self.enable_amp) and dev_type == 'cuda'):
                self.step(data)
            self.post_step()

    def train(self):
        # base_trainer.py:368: torch's per-op profiler ranges (roctx on ROCm) unless cfg.profile_nvtx is off; the profiler state
        # needs a GPU runtime, so a CPU-only process (the host-logic tests) skips it
        with torch.autograd.profiler.emit_nvtx(enabled=bool(self.cfg.profile_nvtx) and torch.cuda.is_available()):
            self.is_optimization_running = True
            while self.is_optimization_running:
                self.iterate()
        return self.return_dict

    def save_model(self):
        from wisp.trainers.validation import save_pipeline
        os.makedirs(self.tracker.log_dir, exist_ok=True)
        name = f'model-ep{self.epoch}-it{self.iteration}.pth' if self.cfg.save_as_new else 'model.pth'
        path = os.path.join(self.tracker.log_dir, name)
        save_pipeline(self.pipeline, path, self.cfg.model_format)
        return path

    # ------------------------------------------------------------------------------------------ events
    def pre_training(self):
        self.tracker.metrics.define_metric('total_loss', aggregation_type=float)

    def post_training(self):
        self.tracker.teardown()

    def pre_epoch(self):
        self.pipeline.train()
        self.tracker.metrics.clear()

    def post_epoch(self):
        self.pipeline.eval()
        self.log_console()
        if self.cfg.save_every > -1 and self.epoch % self.cfg.save_every == 0 and self.epoch != 0:
            self.save_model()

    def pre_step(self):
        pass

    def post_step(self):
        pass

    @abstractmethod
    def step(self, data):
        pass

    @abstractmethod
    def validate(self):
        pass

    def log_console(self):
        log.info('EPOCH {}/{} | total loss: {:>.3E}'.format(self.epoch, self.max_epochs,
                                                           self.tracker.metrics.average_metric('total_loss')))
