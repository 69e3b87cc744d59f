"""Oracle (test infrastructure): sphere tracing of an SDF through octree nuggets, torch-CPU / numpy.
Restates wisp/tracers/packed_sdf_tracer.py:57-174 and wisp/csrc/render/find_depth_bound_cuda.cu:16-45 (including its
bounds quirks: pack i is searched up to the CURRENT index of pack i+1, the last pack up to num_packs).

Parity: PINNED - find_depth_bound bit-exact against the reference kernel body built for the host (tests/golden/depth_bound_ref_*.npz),
sphere_trace against PackedSDFTracer.trace compiled from the reference file and against the reference's whole OctreeGrid / NeuralSDF /
PackedSDFTracer stack on the host - over the unpinned Kaolin leaves (raytrace, pack boundaries)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import spc


def find_depth_bound(query, curr_idxes, depth):
    q = np.asarray(query, dtype=np.float32).reshape(-1)
    cur = np.asarray(curr_idxes, dtype=np.int32)
    P = q.shape[0]
    out = np.full(P, -1, dtype=np.int32)
    for t in range(P):
        if cur[t] > -1:
            i = int(np.uint32(cur[t]))
            stop = P if t == P - 1 else int(np.uint32(cur[t + 1]))
            while i < stop:
                entry, exit_ = depth[i, 0], depth[i, 1]
                if (q[t] >= entry and q[t] <= exit_) or q[t] < entry:
                    out[t] = i
                    break
                i += 1
    return out


def finitediff_gradient(x, f, eps=0.005):
    offs = torch.eye(3) * eps
    return torch.cat([f(x + offs[a]) - f(x - offs[a]) for a in range(3)], dim=-1) / (eps * 2.0)


def sphere_trace(sdf_fn, blas, origins, dirs, dist_max, level, num_steps, step_size, min_dis):
    """Returns dict(xyz, depth, hit, normal, rgb, alpha) per ray (torch float32)."""
    o, d = origins.float(), dirs.float()
    ridx, pidx, depth = spc.raytrace(blas.octree, blas.points, blas.pyramid, blas.exsum, o.numpy(), d.numpy(), level, with_exit=True)
    depth = depth.copy()
    depth[:, 0:1] += np.float32(1e-5)
    first_hit = spc.mark_pack_boundaries(ridx)
    curr = np.nonzero(first_hit)[0].astype(np.int32)
    first_ridx = torch.from_numpy(ridx[first_hit].astype(np.int64))
    nug_o, nug_d = o[first_ridx], d[first_ridx]
    P = first_ridx.shape[0]
    mask = torch.ones(P, dtype=torch.bool)
    hit = torch.zeros(P, dtype=torch.bool)
    depth_t = torch.from_numpy(depth)
    t = depth_t[torch.from_numpy(first_hit)][:, 0:1].clone()
    x = torch.addcmul(nug_o, nug_d, t)
    dist = torch.zeros_like(t)
    curr_t = torch.from_numpy(curr)
    with torch.no_grad():
        if mask.any():
            dist[mask] = sdf_fn(x[mask]) * step_size
        dist[~mask] = 20
        dist_prev = dist.clone()
        for _ in range(num_steps):
            t = t + dist
            x = torch.where(mask.view(-1, 1), torch.addcmul(nug_o, nug_d, t), x)
            hit = torch.where(mask, torch.abs(dist)[..., 0] < min_dis, hit)
            hit = hit | torch.where(mask, torch.abs(dist + dist_prev)[..., 0] * 0.5 < (min_dis * 5), hit)
            mask = torch.where(mask, (t < dist_max)[..., 0], mask)
            mask = mask & ~hit
            if not mask.any():
                break
            dist_prev = torch.where(mask.view(-1, 1), dist, dist_prev)
            nxt = torch.from_numpy(find_depth_bound(t.numpy(), curr_t.numpy(), depth))
            mask = mask & (nxt != -1)
            aabb = nxt != curr_t
            curr_t = torch.where(mask, nxt, curr_t)
            t = torch.where((mask & aabb).view(-1, 1), depth_t[curr_t.long(), 0:1], t)
            x = torch.where(mask.view(-1, 1), torch.addcmul(nug_o, nug_d, t), x)
            if not mask.any():
                break
            dist[mask] = sdf_fn(x[mask]) * step_size
    R = o.shape[0]
    out = dict(xyz=torch.zeros(R, 3), depth=torch.zeros(R, 1), hit=torch.zeros(R, dtype=torch.bool), normal=torch.zeros(R, 3),
               alpha=torch.zeros(R, 1))
    out["hit"][first_ridx] = hit
    out["xyz"][out["hit"]] = x[hit]
    out["depth"][out["hit"]] = t[hit]
    if hit.any():
        out["normal"][out["hit"]] = F.normalize(finitediff_gradient(x[hit], sdf_fn), p=2, dim=-1, eps=1e-5)
    out["rgb"] = (out["normal"] + 1.0) / 2.0
    out["alpha"][out["hit"]] = 1.0
    return out
