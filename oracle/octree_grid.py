"""Oracle (test infrastructure): dual-octree ("trinket") trilinear interpolation and the OctreeGrid /
CodebookOctreeGrid feature lookup, torch-CPU.

Restates kaolin.ops.spc.unbatched_interpolate_trilinear / coords_to_trilinear_coeffs as wisp uses them
(wisp/models/grids/octree_grid.py:130-219, wisp/models/grids/codebook_grid.py:103-172; semantics SURVEY.md A.6).
Kaolin is not available (parity unpinned for this leaf); the half-precision call `feats.half() ... .float()` of
octree_grid.py:147-149 is modelled as: features rounded to fp16, fp32 accumulation, result rounded to fp16.

Parity: the two leaves (interpolate_trilinear, trilinear_coeffs) are UNPINNED; everything built on them - octree_grid_interpolate,
codebook_index_features, codebook_grid_interpolate - is PINNED to OctreeGrid.interpolate / _interpolate and CodebookOctreeGrid._index_features /
_interpolate compiled from the reference files, and to whole-stack renders / training through the reference's classes.
"""
import torch
import torch.nn.functional as F

from . import spc


def trilinear_coeffs(coords, pts, level):
    """coords f32 [..., 3], pts int [..., 3] (voxel origin at `level`) -> [..., 8], j = dx<<2 | dy<<1 | dz."""
    x = (2.0 ** level) * (0.5 * coords.float() + 0.5) - pts.float()
    g = 1.0 - x
    cols = []
    for j in range(8):
        cols.append((x[..., 0] if j & 4 else g[..., 0]) * (x[..., 1] if j & 2 else g[..., 1]) * (x[..., 2] if j & 1 else g[..., 2]))
    return torch.stack(cols, -1)


def interpolate_trilinear(coords, pidx, points, trinkets, feats, level, half_round=False):
    """coords [V,S,3], pidx int [V] (-1 -> zeros), points int16 [P,3], trinkets int32 [P,8], feats [Fn,C] -> [V,S,C]."""
    V, S = coords.shape[:2]
    valid = pidx >= 0
    safe = torch.where(valid, pidx, torch.zeros_like(pidx)).long()
    pts = torch.as_tensor(points)[safe].long()
    w = trilinear_coeffs(coords, pts[:, None, :].expand(V, S, 3), level)                 # [V,S,8]
    f = feats.half().float() if half_round else feats.float()
    corner = f[torch.as_tensor(trinkets)[safe].long()]                                    # [V,8,C]
    acc = None
    for j in range(8):                                                                    # corner order accumulation
        term = corner[:, None, j, :] * w[..., j:j + 1]
        acc = term if acc is None else acc + term
    acc = torch.where(valid[:, None, None], acc, torch.zeros_like(acc))     # pidx == -1 -> exact zeros
    if half_round:
        acc = _ste_half(acc)
    return acc


class _HalfRound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _ste_half(x):
    return _HalfRound.apply(x)


def octree_grid_interpolate(blas, trinkets, features, coords, lod_idx, base_lod, active_lods, multiscale_type,
                            feature_dim, half_round=True):
    """OctreeGrid.interpolate (octree_grid.py:165-219) for coords [B,3]."""
    c = coords.reshape(-1, 3)
    chain = torch.from_numpy(spc.query(blas.octree, blas.exsum, c.detach().numpy(), active_lods[lod_idx], with_parents=True))
    outs = []
    for i in range(lod_idx + 1):
        pidx = chain[:, active_lods[i]]
        outs.append(interpolate_trilinear(c[:, None, :], pidx, blas.points, trinkets, features[i], active_lods[i],
                                          half_round)[:, 0])
    if lod_idx == 0:
        return outs[0]
    feats = torch.cat(outs, -1)
    if multiscale_type == 'sum':
        feats = feats.reshape(c.shape[0], lod_idx + 1, feature_dim).sum(-2)
    return feats


def codebook_index_features(logits, dictionary, training):
    """CodebookOctreeGrid._index_features (codebook_grid.py:103-136): logits [..., K] -> [..., F]."""
    if training:
        y_soft = F.softmax(logits, dim=-1)
        index = y_soft.max(-1, keepdim=True)[1]
        y_hard = torch.zeros_like(logits).scatter_(-1, index, 1.0)
        keys = y_hard - y_soft.detach() + y_soft
        return (dictionary[None, None] * keys[..., None]).sum(-2)
    return dictionary[torch.max(logits, dim=-1)[1]]


def codebook_grid_interpolate(blas, trinkets, features, dictionaries, coords, lod_idx, active_lods, multiscale_type,
                              feature_dim, training):
    """CodebookOctreeGrid through OctreeGrid.interpolate (codebook_grid.py:138-172)."""
    c = coords.reshape(-1, 3)
    chain = torch.from_numpy(spc.query(blas.octree, blas.exsum, c.detach().numpy(), active_lods[lod_idx], with_parents=True))
    outs = []
    for i in range(lod_idx + 1):
        pidx = chain[:, active_lods[i]]
        valid = pidx >= 0
        safe = torch.where(valid, pidx, torch.zeros_like(pidx))
        corner = codebook_index_features(features[i][torch.as_tensor(trinkets)[safe].long()], dictionaries[i], training)  # [V,8,F]
        w = trilinear_coeffs(c, torch.as_tensor(blas.points)[safe].long(), active_lods[i])                                 # [V,8]
        outs.append((corner * w[..., None]).sum(-2) * valid[:, None].float())
    if lod_idx == 0:
        return outs[0]
    feats = torch.cat(outs, -1)
    if multiscale_type == 'sum':
        feats = feats.reshape(c.shape[0], lod_idx + 1, feature_dim).sum(-2)
    return feats
