"""Validation / offline rendering and checkpoint helpers (the caller side right after the hot path).

render()            - OfflineRenderer.render (wisp/trainers/tracker/offline_renderer.py:170-191): chunked no-grad tracing.
evaluate_psnr()     - the PSNR loop of MultiviewTrainer.evaluate_metrics (wisp/trainers/multiview_trainer.py:191-255),
                      returning the same log line the reference prints and its tests scrape (tests/test_utils.py:55-92).
save/load_pipeline  - BaseTrainer.save_model (wisp/trainers/base_trainer.py:344-359).  'state_dict' mode additionally
                      stores the occupancy octree, which the reference silently drops (its OctreeAS tensors are plain
                      attributes, not buffers) so that a pruned model can actually be restored.
"""
import torch

from wisp.core import Rays, RenderBuffer
from wisp.ops.image import psnr


def assert_master_current(module):
    """Raise if a sharded-optimizer trainer has left fp32 table rows of `module` stale on this rank (only their bf16 shadow
    travels between sync_master() calls).  The trainer's guard hangs on the field as a state_dict pre-hook; anything else that
    reads the fp32 master - pickling the whole pipeline, fp32 evaluation - asks here."""
    if not isinstance(module, torch.nn.Module):
        return
    for m in module.modules():
        for hook in getattr(m, "_state_dict_pre_hooks", {}).values():
            fn = getattr(hook, "hook", hook)          # torch wraps hooks in a small record in some versions
            if type(fn).__name__ == "_StaleMasterGuard":
                fn(m, "", False)


def render(pipeline, rays: Rays, lod_idx=None, render_batch=10000, channels=None, amp=False):
    """Full-resolution inference in chunks of `render_batch` rays; RenderBuffer channels concatenated along dim 0."""
    if not amp:
        assert_master_current(pipeline)               # fp32 inference reads the master weights, not the shadow
    kw = {} if channels is None else {"channels": channels}
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        if render_batch > 0:
            rb = RenderBuffer()
            for pack in rays.split(render_batch):
                rb += pipeline.tracer(pipeline.nef, rays=pack, lod_idx=lod_idx, **kw)
            return rb
        return pipeline.tracer(pipeline.nef, rays=rays, lod_idx=lod_idx, **kw)


def evaluate_psnr(pipeline, views, epoch=0, max_epochs=0, name=None, lod_idx=None, render_batch=10000, amp=False):
    """views: iterable of (Rays [H*W,3], rgb [H*W,3]).  Returns (mean psnr, log line)."""
    pipeline.eval()
    total, n = 0.0, 0
    for rays, gts in views:
        rb = render(pipeline, rays.reshape(-1, 3), lod_idx=lod_idx, render_batch=render_batch, channels=["rgb"], amp=amp)
        total += psnr(rb.rgb[..., :3].float(), gts.reshape(-1, 3)[..., :3])
        n += 1
    pipeline.train()
    mean = total / max(n, 1)
    if name is None:
        name = f"lod{pipeline.nef.grid.num_lods - 1}" if lod_idx is None else f"lod{lod_idx}"
    return mean, 'EPOCH {}/{} | {}: {:.2f}'.format(epoch, max_epochs, f"{name} psnr", mean)


BLAS_SIDECAR_SUFFIX = ".blas"


def save_pipeline(pipeline, path, model_format="full"):
    """BaseTrainer.save_model (base_trainer.py:344-359): 'full' pickles the pipeline, anything else writes
    `pipeline.state_dict()` - the bare OrderedDict, byte-compatible with the reference's loaders
    (`pipeline.load_state_dict(torch.load(path))`).  The occupancy octree and the per-cell occupancy record, which the
    reference's state_dict mode silently drops (OctreeAS tensors are plain attributes), go to the sidecar `path + '.blas'`."""
    assert_master_current(pipeline)                   # 'full' pickles the parameters directly: no state_dict hook would fire
    if model_format == "full":
        torch.save(pipeline, path)
        return
    torch.save(pipeline.state_dict(), path)
    blas = getattr(getattr(pipeline.nef, "grid", None), "blas", None)
    if blas is not None:
        occ = getattr(pipeline.nef.grid, "occupancy", None)
        torch.save({"blas_octree": blas.octree.detach().cpu(),
                    "grid_occupancy": None if occ is None else occ.detach().cpu()}, path + BLAS_SIDECAR_SUFFIX)


def _restore_blas(pipeline, extra):
    if extra.get("blas_octree") is None:
        return
    grid = pipeline.nef.grid
    dev = grid.blas.octree.device
    grid.blas = grid.blas.__class__(extra["blas_octree"].to(dev))
    if extra.get("grid_occupancy") is not None:
        grid.occupancy = extra["grid_occupancy"]


def load_pipeline(path, pipeline=None, map_location=None):
    """'full' checkpoints return the pickled pipeline.  A bare state_dict (what the reference and save_pipeline write) is
    loaded into `pipeline`, plus the octree sidecar when one lies next to it; the wrapped {'state_dict': ...} files of
    earlier versions of this package are still read."""
    import os
    from collections.abc import Mapping
    obj = torch.load(path, map_location=map_location, weights_only=False)
    if not isinstance(obj, Mapping):
        return obj
    assert pipeline is not None, "a state_dict checkpoint needs the pipeline to load into"
    if "state_dict" in obj and isinstance(obj["state_dict"], Mapping):
        pipeline.load_state_dict(obj["state_dict"])
        _restore_blas(pipeline, obj)
        return pipeline
    pipeline.load_state_dict(obj)
    side = path + BLAS_SIDECAR_SUFFIX
    if os.path.exists(side):
        _restore_blas(pipeline, torch.load(side, map_location=map_location, weights_only=False))
    return pipeline
