"""Autograd front of the fused NeRF decoder kernel (csrc/nerf_mlp.hip).

Covers NeuralRadianceField.rgba after the grid lookup (wisp/models/nefs/nerf.py:245-264): density MLP, positional
encoding of the view direction, colour MLP, relu / sigmoid.  Parameters are handed to the kernel as one packed fp32
vector in nn.Module order (W1 b1 W2 b2 W3 b3 W4 b4 W5 b5); when the trainer keeps them in one flat buffer
(wisp.trainers.FlatParams) the packed vector and its gradient are zero-copy views of that buffer.
"""
import torch

# decoder shapes the kernels are built for: every app/nerf config of the reference (grid feature width 32 for nerf_hash,
# 5 for nerf_octree / nerf_codebook, 12 for nerf_triplanar; hidden 64, one hidden layer, 4 view octaves)
SUPPORTED = dict(max_in_dim=32, hidden=64, view_freqs=4)
# hidden widths with a fused kernel: 64 (register-chained forward AND weight gradients, fp32 or bf16 compute) and 128 (the
# reference's best nerf_hash row / the documented VQAD command line: csrc/nerf_mlp_wide.hip, bf16 compute only)
FUSED_HIDDEN = (64, 128)
BF16_ONLY_HIDDEN = (128,)


def _compute_bf16(nef):
    mode = getattr(nef, 'decoder_compute', 'auto')
    return (mode == 'bf16') or (mode == 'auto' and torch.is_autocast_enabled())


def _hip():
    import wisp._C as _C
    return _C


def _decoder_tensors(nef):
    dd, dc = nef.decoder_density, nef.decoder_color
    layers = [dd.layers[0], dd.lout, dc.layers[0], dc.layers[1], dc.lout]
    out = []
    for l in layers:
        out.append(l.weight)
        out.append(l.bias)          # may be None (bias=False configs)
    return out


def _flat_view(tensors):
    """One contiguous fp32 view covering `tensors` if they sit back to back in the same storage, else None."""
    if any(t is None for t in tensors):
        return None
    first = tensors[0]
    ptr, total = first.data_ptr(), 0
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != ptr + 4 * total:
            return None
        total += t.numel()
    return torch.empty(0, dtype=torch.float32, device=first.device).set_(first.untyped_storage(), first.storage_offset(),
                                                                         (total,), (1,))


def _pack(tensors, shapes):
    parts = []
    for t, shp in zip(tensors, shapes):
        parts.append(torch.zeros(shp, dtype=torch.float32, device=tensors[0].device).reshape(-1) if t is None
                     else t.detach().reshape(-1).float())
    return torch.cat(parts)


class _FusedDecoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, dirs, compute_bf16, shapes, *params):
        C = _hip()
        flat = _flat_view([p.detach() if p is not None else None for p in params])
        packed = flat if flat is not None else _pack(params, shapes)
        hidden = shapes[0][0]
        rgb, density = C.nerf_mlp_forward(feats.detach(), dirs, packed, feats.shape[-1], hidden, SUPPORTED["view_freqs"],
                                          compute_bf16)
        ctx.save_for_backward(feats.detach(), dirs, packed)
        ctx.compute_bf16, ctx.shapes = compute_bf16, shapes
        # in-place gradient accumulation when the parameters' .grad tensors are views of one flat buffer
        grads = [getattr(p, 'grad', None) if p is not None else None for p in params]
        ctx.grad_flat = _flat_view(grads) if (flat is not None and all(g is not None for g in grads)) else None
        ctx.present = [p is not None for p in params]
        return rgb, density

    @staticmethod
    def backward(ctx, g_rgb, g_density):
        C = _hip()
        feats, dirs, packed = ctx.saved_tensors
        if g_rgb is None:
            g_rgb = torch.zeros(feats.shape[0], 3, device=feats.device)
        if g_density is None:
            g_density = torch.zeros(feats.shape[0], 1, device=feats.device)
        g_feats, g_params = C.nerf_mlp_backward(feats, dirs, packed, g_rgb.contiguous().float(), g_density.contiguous().float(),
                                                feats.shape[-1], ctx.shapes[0][0], SUPPORTED["view_freqs"],
                                                ctx.compute_bf16, grad_params=ctx.grad_flat)
        if ctx.grad_flat is not None:
            return (g_feats, None, None, None) + tuple(None for _ in ctx.present)
        outs, off = [], 0
        for shp, present in zip(ctx.shapes, ctx.present):
            n = 1
            for s in shp:
                n *= s
            outs.append(g_params[off:off + n].reshape(shp) if present else None)
            off += n
        return (g_feats, None, None, None) + tuple(outs)


def supports(nef, feats):
    if nef.hidden_dim not in FUSED_HIDDEN or (nef.hidden_dim in BF16_ONLY_HIDDEN and not _compute_bf16(nef)):
        return False
    return (feats.is_cuda and 1 <= feats.shape[-1] <= SUPPORTED["max_in_dim"]
            and nef.view_multires == SUPPORTED["view_freqs"] and nef.num_layers == 1 and nef.pos_embedder is None
            and nef.view_embedder_type == 'positional' and nef.activation_type == 'relu'
            and nef.layer_type in ('linear', 'none') and feats.dtype in (torch.float32, torch.float16, torch.bfloat16))


def fused_nerf_decoder(nef, feats, ray_d):
    """(rgb [S,3] fp32, density [S,1] fp32) for the decoder shapes of the reference's app/nerf configs (see SUPPORTED)."""
    compute_bf16 = _compute_bf16(nef)
    params = _decoder_tensors(nef)
    H, I = nef.hidden_dim, feats.shape[-1]
    shapes = ((H, I), (H,), (16, H), (16,), (H, 42), (H,), (H, H), (H,), (3, H), (3,))
    return _FusedDecoder.apply(feats.contiguous(), ray_d.contiguous().float(), compute_bf16, shapes, *params)
