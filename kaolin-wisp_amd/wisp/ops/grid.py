"""Grid interpolation ops: autograd front of the HIP hash-grid kernels.

Counterpart of wisp/ops/grid.py:16-144.  `hashgrid` / `HashGridInterpolate` keep the reference signature;
forward and backward are single launches over all levels (csrc/hashgrid.hip).  Differences: the gradient table
is accumulated in fp32 (the reference adds __half2 atomics in the table dtype), bf16 tables are accepted, and
the 'cat' zeroing of HashGrid.interpolate can be fused through `zero_from_col`.
"""
import weakref

import torch

PRIMES = [1, 2654435761, 805459861]


def _hip():
    import wisp._C as _C
    return _C


def _as_int_list(resolutions):
    if torch.is_tensor(resolutions):
        return [int(r) for r in resolutions.reshape(-1).tolist()]
    return [int(r) for r in resolutions]


def hashgrid_naive(coords, resolutions, codebook_bitwidth, lod_idx, codebook, codebook_lod_sizes, codebook_lod_first_idx):
    """Plain-PyTorch sketch of the hash-grid lookup, signature of wisp/ops/grid.py:16-76 (levels 0..lod_idx, [batch, (lod_idx+1)*F]).
    Like the reference's, it is documentation, not the product path and not numerically 1:1 with the kernels (float32
    scaling with torch.clip, `res**3 > T` as the hash test, int64 hashing: SURVEY 3.4-3); nothing in this package calls it.

    Args:
        coords (torch.FloatTensor): [batch, 3] in [-1, 1]
        resolutions: per-level grid resolution, [num_lods]
        codebook_bitwidth (int): hashed levels have 2^bitwidth rows
        lod_idx (int): last level to evaluate
        codebook (torch.Tensor): stacked per-level tables [sum rows, F]
        codebook_lod_sizes / codebook_lod_first_idx: rows and first row of every level
    """
    size = 2 ** codebook_bitwidth
    pts = coords.reshape(-1, 3)
    offs = torch.tensor([[(j >> 2) & 1, (j >> 1) & 1, j & 1] for j in range(8)], device=pts.device)     # corner j = (dx, dy, dz)
    out = []
    for i, res in enumerate(_as_int_list(resolutions)[:lod_idx + 1]):
        x = torch.clip((pts + 1.0) * 0.5 * res, 0, res - 1 - 1e-5)
        base = torch.floor(x)
        frac = x - base
        corner = base.long()[:, None, :] + offs[None]                                  # [B, 8, 3]
        if res ** 3 > size:
            idx = ((corner[..., 0] * PRIMES[0]) ^ (corner[..., 1] * PRIMES[1]) ^ (corner[..., 2] * PRIMES[2])) % size
        else:
            idx = corner[..., 0] + corner[..., 1] * res + corner[..., 2] * res * res
        first = int(codebook_lod_first_idx[i])
        table = codebook[first:first + int(codebook_lod_sizes[i])]
        w = torch.where(offs[None].bool(), frac[:, None, :], 1.0 - frac[:, None, :]).prod(-1)          # [B, 8]
        out.append((table[idx] * w[..., None].to(table.dtype)).sum(1))
    return torch.cat(out, -1)


# Trainer-side helpers of a table Parameter, kept OUT of Parameter.__dict__ (which torch.save(pipeline) would pickle: the
# flat gradient / shadow storages would travel with the checkpoint and come back dangling).  Weak keys: an entry dies
# with its Parameter.
class _ById:
    """id-keyed weak registry (a WeakKeyDictionary would compare keys with Tensor.__eq__, which is elementwise)."""

    def __init__(self):
        self._d = {}

    def get(self, obj):
        ent = self._d.get(id(obj))
        return ent[1] if ent is not None and ent[0]() is obj else None

    def __setitem__(self, obj, value):
        key = id(obj)
        self._d[key] = (weakref.ref(obj, lambda _r, k=key, d=self._d: d.pop(k, None)), value)


_TABLE_AUX = _ById()


class _TableAux:
    __slots__ = ("shadow", "shadow_version", "grad_buffer")

    def __init__(self):
        self.shadow, self.shadow_version, self.grad_buffer = None, -1, None


def register_table_aux(codebook, shadow=None, grad_buffer=None):
    """A trainer announces (a) a low-precision copy of `codebook` that its fused optimizer keeps current and/or (b) the
    pre-allocated fp32 gradient buffer the backward may scatter into.  Both are validated at every use."""
    aux = _TABLE_AUX.get(codebook)
    if aux is None:
        aux = _TABLE_AUX[codebook] = _TableAux()
    if shadow is not None:
        aux.shadow, aux.shadow_version = shadow, codebook._version
    if grad_buffer is not None:
        aux.grad_buffer = grad_buffer
    return aux


def mark_shadow_current(codebook):
    """Called by the trainer right after its fused optimizer rewrote table AND shadow (the raw-pointer kernels do not bump
    the autograd version counter; any torch-side in-place write - load_state_dict, .copy_, an external optimizer - does,
    which is what invalidates the shadow until the next fused step)."""
    aux = _TABLE_AUX.get(codebook)
    if aux is not None and aux.shadow is not None:
        aux.shadow_version = codebook._version


def current_shadow(codebook, dtype):
    """The registered copy of `codebook` in `dtype` if it is known to mirror the master weights, else None."""
    aux = _TABLE_AUX.get(codebook)
    if aux is None or aux.shadow is None or aux.shadow.dtype != dtype or aux.shadow_version != codebook._version:
        return None
    return aux.shadow


def current_grad_buffer(codebook):
    """The registered gradient buffer iff it IS codebook.grad (same storage) - otherwise autograd gets a normal gradient."""
    aux = _TABLE_AUX.get(codebook)
    if aux is None or aux.grad_buffer is None or codebook.grad is None:
        return None
    buf = aux.grad_buffer
    if codebook.grad.data_ptr() != buf.data_ptr() or buf.dtype != torch.float32 or tuple(buf.shape) != tuple(codebook.shape):
        return None
    return buf


class HashGridInterpolate(torch.autograd.Function):
    """feats[N, L*F] = multi-resolution (dense or hashed) trilinear / bilinear lookup of `codebook`."""

    @staticmethod
    def forward(ctx, coords, resolutions, codebook_bitwidth, lod_idx, codebook, codebook_first_idx, zero_from_col=None):
        if codebook.shape[-1] % 2 == 1:
            raise Exception("The codebook feature dimension needs to be a multiple of 2.")
        assert coords.shape[-1] in [2, 3]
        table = codebook
        if torch.is_autocast_enabled():
            # the reference casts to fp16 under autocast (grid.py:88-89); follow the active autocast dtype (bf16 on MI355X).
            # A trainer may keep an up-to-date low-precision copy next to the master weights (refreshed by the fused AdamW).
            dt = torch.get_autocast_dtype('cuda')
            shadow = current_shadow(codebook, dt)
            table = shadow if shadow is not None else codebook.to(dt)
        res = _as_int_list(resolutions)
        feats = _hip().hashgrid_interpolate(coords.detach(), table.detach(), codebook_first_idx, res, codebook_bitwidth,
                                            zero_from_col)
        # (grad w.r.t. coords reads the table the forward read - the low-precision copy under autocast - so it is saved with them)
        if coords.requires_grad:
            ctx.save_for_backward(coords, codebook_first_idx, table.detach())
        else:
            ctx.save_for_backward(coords, codebook_first_idx)
        ctx.meta = (res, codebook_bitwidth, tuple(codebook.shape), codebook.dtype, zero_from_col)
        # a trainer may pre-allocate the fp32 gradient buffer of the table (flat-parameter layout): scatter into it
        ctx.grad_buffer = current_grad_buffer(codebook)
        return feats

    @staticmethod
    def backward(ctx, grad_output):
        coords, first_idx = ctx.saved_tensors[:2]
        res, bitwidth, shape, dtype, zero_from_col = ctx.meta
        grad_coords = None
        if ctx.needs_input_grad[0]:
            # what the reference returns for coords that require a gradient (grid.py:109-126 -> hashgrid_interpolate_cuda.cu:163-196),
            # its arithmetic as is - see wisp_hashgrid_grad_coords in include/wisp_hip.h for what that arithmetic is and is not
            if len(ctx.saved_tensors) < 3:
                raise RuntimeError("HashGridInterpolate: coords did not require a gradient in forward, but one is asked for now")
            grad_coords = _hip().hashgrid_grad_coords(coords.detach().float(), grad_output.contiguous(), ctx.saved_tensors[2], first_idx,
                                                      res, bitwidth)
            if coords.shape[-1] != 3:
                grad_coords = grad_coords[:, :coords.shape[-1]]           # (2-D: the reference's [n, 3] zeros do not fit [n, 2] coords)
        buf = ctx.grad_buffer
        grad = _hip().hashgrid_interpolate_backward(coords.detach().float(), grad_output.contiguous(), shape, first_idx,
                                                    res, bitwidth, zero_from_col, out=buf)
        if buf is not None:
            return grad_coords, None, None, None, None, None, None      # accumulated in place, nothing for autograd to add
        return grad_coords, None, None, None, grad.to(dtype), None, None


def hashgrid(coords, codebook_bitwidth, lod_idx, codebook, zero_from_col=None):
    """Hash-grid query + interpolation.

    Args:
        coords (torch.FloatTensor): [batch, 2 or 3] in [-1, 1]
        codebook_bitwidth (int): the hashed levels have 2^bitwidth entries
        lod_idx (int): unused by the kernel (all levels are evaluated, as in the reference)
        codebook (wisp.models.grids.utils.MultiTable): stacked per-level tables
    Returns:
        (torch.Tensor): [batch, num_lods * feature_dim]
    """
    batch, dim = coords.shape
    feats = HashGridInterpolate.apply(coords.contiguous(), codebook.resolutions, codebook_bitwidth, lod_idx,
                                      codebook.feats, codebook.begin_idxes, zero_from_col)
    feature_dim = codebook.feats.shape[1] * len(codebook.resolutions)
    return feats.reshape(batch, feature_dim)


class GridInterpolate(torch.autograd.Function):
    """Trilinear blend of eight gathered corner rows (wisp/ops/grid.py:146-168): feats [N,8,F], local coords [N,3] in [0,1]."""

    @staticmethod
    def forward(ctx, coords, feats):
        if torch.is_autocast_enabled():
            feats = feats.float()                                  # custom_fwd(cast_inputs=torch.float) in the reference
        out = _hip().ops.grid_interpolate_cuda(coords.float().contiguous(), feats.contiguous())
        ctx.save_for_backward(coords)
        ctx.feature_dim = feats.shape[-1]
        return out

    @staticmethod
    def backward(ctx, grad_output):
        coords = ctx.saved_tensors[0]
        return None, _hip().ops.grid_interpolate_backward_cuda(coords.float().contiguous(), grad_output.contiguous(),
                                                               ctx.feature_dim)


def grid_interpolate(coords, feats):
    """feats [N,8,F] blended with the trilinear weights of the local coordinates [N,3] -> [N,F]."""
    return GridInterpolate.apply(coords.contiguous(), feats.contiguous())


class HashGridQuery(torch.autograd.Function):
    """The eight un-blended corner rows of every level (wisp/ops/grid.py:169-209): differentiable w.r.t. the per-level
    codebooks.  Under autocast the reference casts its inputs to half (custom_fwd(cast_inputs=torch.half))."""

    @staticmethod
    def forward(ctx, coords, resolutions, codebook_bitwidth, probe_bitwidth, lod_idx, *codebook):
        if codebook[0].shape[-1] % 2 == 1:
            raise Exception("The codebook feature dimension needs to be a multiple of 2.")
        if torch.is_autocast_enabled():
            codebook = tuple(c.half() for c in codebook)
        feats = _hip().ops.hashgrid_query_cuda(coords.float().contiguous(), [c.contiguous() for c in codebook],
                                               resolutions, codebook_bitwidth, probe_bitwidth).contiguous()
        ctx.save_for_backward(coords)
        ctx.resolutions = resolutions
        ctx.codebook_rows = [c.shape[0] for c in codebook]
        ctx.codebook_dtypes = [c.dtype for c in codebook]
        ctx.codebook_bitwidth = codebook_bitwidth
        ctx.feature_dim = codebook[0].shape[-1]
        ctx.probe_bitwidth = probe_bitwidth
        return feats

    @staticmethod
    def backward(ctx, grad_output):
        coords = ctx.saved_tensors[0]
        grads = _hip().ops.hashgrid_query_backward_cuda(coords.float().contiguous(), grad_output.contiguous(), ctx.resolutions,
                                                        ctx.codebook_rows, ctx.codebook_bitwidth, ctx.feature_dim,
                                                        ctx.probe_bitwidth)
        return (None, None, None, None, None, *grads)


def hashgrid_query_fwd(coords, resolutions, codebook_bitwidth, lod_idx, codebook, probe_bitwidth=0):
    """Non-differentiable corner query (wisp/ops/grid.py:211-224) -> [batch, 8, feature_dim * num_lods * 2^probe_bitwidth]."""
    batch, dim = coords.shape
    assert coords.shape[-1] in [2, 3]
    feats = _hip().ops.hashgrid_query_cuda(coords.float().contiguous(), [c.contiguous() for c in codebook], resolutions,
                                           codebook_bitwidth, probe_bitwidth).contiguous()
    feature_dim = codebook[0].shape[1] * len(resolutions)
    return feats.reshape(batch, 8, feature_dim * (2 ** probe_bitwidth))


def hashgrid_query(coords, resolutions, codebook_bitwidth, lod_idx, codebook, probe_bitwidth=0):
    """Differentiable corner query (wisp/ops/grid.py:226-245): coords [batch, 3], codebook = one [2^bw, feature_dim] tensor per
    level -> [batch, 8, feature_dim * num_lods * 2^probe_bitwidth].  (`lod_idx` is accepted and ignored, as in the reference.)"""
    batch, dim = coords.shape
    assert coords.shape[-1] in [2, 3]
    feats = HashGridQuery.apply(coords.contiguous(), resolutions, codebook_bitwidth, probe_bitwidth, lod_idx, *[c for c in codebook])
    feature_dim = codebook[0].shape[1] * len(resolutions)
    return feats.reshape(batch, 8, feature_dim * (2 ** probe_bitwidth))


class SPCTrilinear(torch.autograd.Function):
    """Differentiable (w.r.t. the features) dual-octree trilinear interpolation - the Kaolin-Core leaf
    unbatched_interpolate_trilinear that OctreeGrid._interpolate calls (wisp/models/grids/octree_grid.py:147-149)."""

    @staticmethod
    def forward(ctx, coords, pidx, points, trinkets, feats, level, half_round):
        out = _hip().spc_trilinear_forward(coords.detach(), pidx, points, trinkets, feats.detach(), level, half_round)
        ctx.save_for_backward(coords.detach(), pidx, points, trinkets)
        ctx.meta = (tuple(feats.shape), feats.dtype, level)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coords, pidx, points, trinkets = ctx.saved_tensors
        shape, dtype, level = ctx.meta
        grad = _hip().spc_trilinear_backward(coords, pidx, points, trinkets, grad_out.contiguous().float(), shape, level)
        return None, None, None, None, grad.to(dtype), None, None


def spc_interpolate_trilinear(coords, pidx, points, trinkets, feats, level, half_round=False):
    """coords [V,S,3], pidx [V] (-1 = outside), feats [Fn,C] -> [V,S,C] float32."""
    return SPCTrilinear.apply(coords.contiguous(), pidx, points, trinkets, feats, level, half_round)


class SPCTrilinearMulti(torch.autograd.Function):
    """Every active level of an OctreeGrid at once (octree_grid.py:183-219): per-level trilinear lookups written straight
    into the concatenated row or summed, one launch forward and one backward."""

    @staticmethod
    def forward(ctx, coords, chain, points, trinkets, levels, half_round, sum_lods, *feats):
        out = _hip().spc_trilinear_multi_forward(coords.detach(), chain, points, trinkets, [f.detach() for f in feats], levels,
                                                 half_round, sum_lods)
        ctx.save_for_backward(coords.detach(), chain, points, trinkets)
        ctx.meta = ([tuple(f.shape) for f in feats], [f.dtype for f in feats], list(levels), sum_lods)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coords, chain, points, trinkets = ctx.saved_tensors
        shapes, dtypes, levels, sum_lods = ctx.meta
        grads = _hip().spc_trilinear_multi_backward(coords, chain, points, trinkets, grad_out.contiguous().float(), shapes,
                                                    levels, sum_lods)
        return (None,) * 7 + tuple(g.to(dt) for g, dt in zip(grads, dtypes))


def spc_interpolate_trilinear_multi(coords, chain, points, trinkets, feats, levels, half_round=False, sum_lods=False):
    """coords [N,3], chain i64 [N, >= len(feats)] (voxel per level, -1 = outside) -> [N, L*C] or (sum_lods) [N, C]."""
    return SPCTrilinearMulti.apply(coords.contiguous(), chain, points, trinkets, tuple(int(l) for l in levels), half_round,
                                   sum_lods, *feats)


def coords_to_trilinear_coeffs(coords, points, level):
    """coords [V,S,3] + the quantised voxel origin of every row ([V,3] or [V,S,3] repeated) -> [V,S,8]."""
    pts = points[:, 0] if points.ndim == 3 else points
    return _hip().spc_trilinear_coeffs(coords.contiguous(), pts.contiguous(), level)


class CodebookTrilinear(torch.autograd.Function):
    """Fused VQAD dictionary selection (straight-through softmax one-hot in training, argmax in eval) + trilinear blend;
    what CodebookOctreeGrid._index_features / _interpolate compute (wisp/models/grids/codebook_grid.py:103-172)."""

    @staticmethod
    def forward(ctx, coords, pidx, points, trinkets, logits, dictionary, level, training):
        out = _hip().codebook_trilinear_forward(coords.detach(), pidx, points, trinkets, logits.detach(), dictionary.detach(),
                                                level, training)
        ctx.save_for_backward(coords.detach(), pidx, points, trinkets, logits.detach(), dictionary.detach())
        ctx.level, ctx.training = level, training
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coords, pidx, points, trinkets, logits, dictionary = ctx.saved_tensors
        if not ctx.training:
            return (None,) * 8          # eval-mode lookup is a hard argmax: no gradient path (as in the reference)
        gl, gd = _hip().codebook_trilinear_backward(coords, pidx, points, trinkets, logits, dictionary,
                                                    grad_out.contiguous().float(), ctx.level)
        return None, None, None, None, gl, gd, None, None


def codebook_interpolate_trilinear(coords, pidx, points, trinkets, logits, dictionary, level, training):
    """coords [V,S,3], pidx [V] -> [V,S,F] float32."""
    return CodebookTrilinear.apply(coords.contiguous(), pidx, points, trinkets, logits, dictionary, level, training)
