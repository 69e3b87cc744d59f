"""How much of "bit-exact ridx / boundary" depends on float orderings that Kaolin's absent source would decide?

The oracle (and the HIP kernels) fix three orderings inside the Kaolin leaves / torch ops of OctreeAS._raymarch_ray
(octree_as.py:272-300): the sample position o + fl(d*t), the query cell floor(2^L * fl(0.5 x + 0.5)) and "inside iff |x| <= 1".
This script counts, at the flagship shape (SynLego level-7 occupancy, 16 384 rays x 2048 candidates = 33.5 M), the candidates
whose integer cell / whose occupancy decision changes under each alternative a different implementation could have chosen.
CPU only (numpy); run from the repo root:  python scripts/float_order_bound.py [rays]
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaolin-wisp_amd"))
import synlego                                    # noqa: E402
from oracle import raymarch as omarch             # noqa: E402

F32 = np.float32
LEVEL, N = 7, 2048
RES = 2 ** LEVEL


def cell_q0(x):                     # the oracle / kernels: floor(res * fl(0.5 x + 0.5)), clamped
    return np.minimum(np.floor(F32(RES) * (F32(0.5) * x + F32(0.5))), RES - 1).astype(np.int32)


def cell_q1(x):                     # 0.5f * (x + 1) * res, left to right
    return np.minimum(np.floor(((F32(0.5) * (x + F32(1.0)).astype(F32)).astype(F32) * F32(RES)).astype(F32)), RES - 1).astype(np.int32)


def cell_q2(x):                     # (x + 1) * (res / 2)
    return np.minimum(np.floor(((x + F32(1.0)).astype(F32) * F32(RES / 2)).astype(F32)), RES - 1).astype(np.int32)


def cell_q3(x):                     # double arithmetic, rounded once: float(res * (x * 0.5 + 0.5))  (the hash grid's style)
    return np.minimum(np.floor((RES * (x.astype(np.float64) * 0.5 + 0.5)).astype(F32)), RES - 1).astype(np.int32)


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    import torch
    cells = synlego.occupied_cells(LEVEL).cpu().numpy().astype(np.int64)
    occ = np.zeros((RES, RES, RES), dtype=bool)
    occ[cells[:, 0], cells[:, 1], cells[:, 2]] = True
    o, d, _ = synlego.ray_bank(R, seed=321, device='cpu', with_gt=False)
    o, d = o.numpy().astype(F32), d.numpy().astype(F32)
    rng = np.random.default_rng(5)
    tot = 0
    keys = ["pos_fma: cell", "pos_fma: occupancy", "q1 0.5f*(x+1)*res: cell", "q1: occupancy", "q2 (x+1)*(res/2): cell", "q2: occupancy",
            "q3 double, one rounding: cell", "q3: occupancy", "x == +-1 outside: occupancy", "any |x| == 1 exactly"]
    cnt = dict.fromkeys(keys, 0)
    kept = 0
    for s in range(0, R, 1024):
        oo, dd = o[s:s + 1024], d[s:s + 1024]
        r = oo.shape[0]
        depth = omarch.ray_depths(r, N, synlego.NEAR, synlego.FAR, rng.uniform(size=(r, N)).astype(F32))
        p0 = (oo[:, None, :] + (dd[:, None, :] * depth[:, :, None]).astype(F32)).astype(F32).reshape(-1, 3)
        # fused multiply-add: exact product (48 bits fit a double), the sum rounded to double and then to float - differs from a
        # hardware fma only when the double sum is itself inexact AND lands on a float tie (< 1e-8 of inputs)
        p1 = (oo[:, None, :].astype(np.float64) + dd[:, None, :].astype(np.float64) * depth[:, :, None].astype(np.float64)).astype(F32).reshape(-1, 3)
        inside0 = np.all(np.abs(p0) <= F32(1.0), axis=1)

        def occupied(cell, inside):
            c = np.clip(cell, 0, RES - 1)
            return inside & occ[c[:, 0], c[:, 1], c[:, 2]]
        c0 = cell_q0(p0)
        base = occupied(c0, inside0)
        kept += int(base.sum())
        tot += p0.shape[0]
        c = cell_q0(p1)
        ins1 = np.all(np.abs(p1) <= F32(1.0), axis=1)
        cnt["pos_fma: cell"] += int(((c != c0).any(axis=1) & inside0 & ins1).sum())
        cnt["pos_fma: occupancy"] += int((occupied(c, ins1) != base).sum())
        for name, fn in (("q1 0.5f*(x+1)*res", cell_q1), ("q2 (x+1)*(res/2)", cell_q2), ("q3 double, one rounding", cell_q3)):
            c = fn(p0)
            cnt[name + ": cell"] += int(((c != c0).any(axis=1) & inside0).sum())
            cnt[name.split(" ")[0] + ": occupancy"] += int((occupied(c, inside0) != base).sum())
        strict = np.all(np.abs(p0) < F32(1.0), axis=1)
        cnt["x == +-1 outside: occupancy"] += int((occupied(c0, strict) != base).sum())
        cnt["any |x| == 1 exactly"] += int((inside0 & ~strict).sum())
    print(f"flagship shape: {R} rays x {N} candidates = {tot} candidates, {kept} inside occupied cells "
          f"({len(cells)} of {RES ** 3} level-{LEVEL} cells occupied)")
    print(f"{'alternative ordering':44s} {'candidates that change':>24s} {'rate':>12s}")
    for k in keys:
        print(f"{k:44s} {cnt[k]:24d} {cnt[k] / tot:12.3e}")


if __name__ == "__main__":
    main()
