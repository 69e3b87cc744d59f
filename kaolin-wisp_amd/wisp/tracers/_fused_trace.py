"""One autograd node for PackedRFTracer.trace's differentiable part when the field is the shipped NeRF shape.

An unchanged application (wisp.trainers.MultiviewTrainer, app/nerf/main_nerf.py) reaches the hot path through
Pipeline.forward -> BaseTracer.forward -> PackedRFTracer.trace -> BaseNeuralField.forward -> NeuralRadianceField.rgba, i.e.
through three custom autograd Functions (hash-grid lookup, fused decoder, fused compositing) and the Python between them.
At the reference trainer's 2^18 samples per step that Python - module dispatch by introspection, three Function.apply round
trips, reshape / index_select nodes, the engine walking the graph backwards - is longer than the GPU work.  Here the same three
Functions' own `forward` / `backward` bodies run back to back inside ONE Function (a stand-in context object each), so the
arithmetic, dtypes, autocast behaviour and in-place gradient buffers are exactly the modular path's - the tests compare the two
bit for bit - while autograd sees one node with the hash table and the ten decoder tensors as its inputs.

WISP_FUSED_TRACE=0 keeps the modular path."""
import os

import torch

from wisp.ops.grid import HashGridInterpolate
from wisp.ops.nerf_mlp import _FusedDecoder, _compute_bf16, _decoder_tensors, supports as decoder_supports
from wisp.ops.render import _Composite

ENABLED = os.environ.get("WISP_FUSED_TRACE", "1") != "0"


class _Ctx:
    """What the three Functions use of an autograd context."""
    needs_input_grad = (False,) * 32

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass


def supports(nef, lod_idx, extra_channels):
    """The nerf_hash.yaml shape: NeuralRadianceField over a 'cat' HashGrid queried at its finest LOD, decoder shape of the fused
    kernels, nothing but rgb / alpha / depth / hit asked for."""
    if not ENABLED or extra_channels or not torch.is_grad_enabled():
        return False
    from wisp.models.grids import HashGrid
    from wisp.models.nefs import NeuralRadianceField
    grid = getattr(nef, "grid", None)
    if type(nef) is not NeuralRadianceField or type(grid) is not HashGrid or grid.multiscale_type != 'cat':
        return False
    if not getattr(nef, "fused_decoder", False) or getattr(nef, "pos_embedder", None) is not None:
        return False
    feats = grid.codebook.feats
    if not feats.is_cuda or feats.shape[-1] % 2 == 1:
        return False
    probe = torch.empty(0, nef.effective_feature_dim(), dtype=feats.dtype, device=feats.device)
    return decoder_supports(nef, probe)


class _FusedTrace(torch.autograd.Function):
    @staticmethod
    def forward(ctx, st, table, *dec):
        """st: dict of everything that is not a parameter (samples, view directions, spacing, offsets, static shapes).
        table + dec: the tensors autograd differentiates for - exactly the inputs the three inner Functions take."""
        nef, grid = st["nef"], st["nef"].grid
        S = st["samples"].shape[0]
        g, d, c = _Ctx(), _Ctx(), _Ctx()
        feats = HashGridInterpolate.forward(g, st["samples"], grid.codebook.resolutions, grid.codebook_bitwidth, st["lod_idx"], table,
                                            grid.codebook.begin_idxes, st["lod_idx"] * grid.feature_dim)
        feats = feats.reshape(S, nef.effective_feature_dim())
        H, I = nef.hidden_dim, feats.shape[-1]
        shapes = ((H, I), (H,), (16, H), (16,), (H, 42), (H,), (H, H), (H,), (3, H), (3,))
        color, density = _FusedDecoder.forward(d, feats.contiguous(), st["dirs"], _compute_bf16(nef), shapes, *dec)
        rgb, alpha, depth, hit = _Composite.forward(c, color, density.reshape(S, 1), st["deltas"], st["depths"], None, st["offsets"],
                                                    st["num_rays"], st["bg"])
        ctx.inner = (g, d, c)
        ctx.mark_non_differentiable(hit)
        return rgb, alpha, depth, hit

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth, _g_hit):
        g, d, c = ctx.inner
        gc, gd = _Composite.backward(c, g_rgb, g_alpha, g_depth, None)[:2]
        out = _FusedDecoder.backward(d, gc, gd)
        g_table = HashGridInterpolate.backward(g, out[0])[4]
        return (None, g_table) + tuple(out[4:])


def fused_trace(nef, samples, dirs, deltas, depths, offsets, num_rays, bg, lod_idx):
    """-> (rgb [R,3], alpha [R,1], depth [R,1] or None, hit bool [R]) like wisp.ops.render.composite over nef.rgba's outputs."""
    if samples.requires_grad or dirs.requires_grad:
        raise NotImplementedError("fused_trace differentiates the table and the decoder only; coordinates / directions that require a "
                                  "gradient go through the modular path (PackedRFTracer.trace picks it)")
    st = dict(nef=nef, samples=samples.contiguous(), dirs=dirs.contiguous().float(), deltas=deltas, depths=depths, offsets=offsets,
              num_rays=num_rays, bg=bg, lod_idx=lod_idx)
    rgb, alpha, depth, hit = _FusedTrace.apply(st, nef.grid.codebook.feats, *_decoder_tensors(nef))
    return rgb, alpha, (depth if depths is not None else None), hit
