"""Adversarial coordinate sets for the hash-grid cell arithmetic (test infrastructure, shared by the CPU and GPU suites).

The reference scales a coordinate in double - float(res * (c * 0.5 + 0.5)), hashgrid_interpolate_cuda.cu:40-42 - this package's
kernels with ONE fp32 fma (csrc/hashgrid.hip corner_setup).  The two can only differ where the double sum is inexact
(|c| < 2^-18) and lands on a float rounding midpoint, and a difference only matters where it moves floor(x): at cell faces.
So the set concentrates there."""
import numpy as np

NGP_RES = [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512]


def _ulp_neighbours(v, k=3):
    """v (float32 array) and its k nearest floats on either side."""
    v = np.asarray(v, dtype=np.float32)
    out = [v]
    up, dn = v.copy(), v.copy()
    for _ in range(k):
        up = np.nextafter(up, np.float32(np.inf))
        dn = np.nextafter(dn, np.float32(-np.inf))
        out += [up.copy(), dn.copy()]
    return np.concatenate(out)


def structured_scalars(resolutions=NGP_RES):
    """Every cell face c = 2k/res - 1 of every level (rounded to float32 both ways) +-3 ulp, the powers of two 2^-149..2^1 of
    either sign +-3 ulp, zero / one / the float limits of the denormal range."""
    parts = []
    for res in resolutions:
        k = np.arange(res + 1, dtype=np.float64)
        face = 2.0 * k / res - 1.0
        parts.append(_ulp_neighbours(face.astype(np.float32)))
        # the coordinate whose DOUBLE image sits exactly on the face may differ from the rounded one: add the half-cell too
        parts.append(_ulp_neighbours(((2.0 * k + 1.0) / res - 1.0).astype(np.float32), 1))
    p2 = np.ldexp(1.0, np.arange(-149, 2)).astype(np.float32)
    parts.append(_ulp_neighbours(np.concatenate([p2, -p2])))
    tiny = np.float32(np.finfo(np.float32).tiny)
    parts.append(np.array([0.0, -0.0, 1.0, -1.0, tiny, -tiny, np.nextafter(tiny, np.float32(0)), 1.5, -1.5, 4.0, -4.0,
                           np.float32(1.0) + np.float32(2 ** -23), -(np.float32(1.0) + np.float32(2 ** -23))], dtype=np.float32))
    return np.unique(np.concatenate(parts).view(np.uint32)).view(np.float32)


def random_scalars(n_tiny, n_unit, n_out, seed):
    rng = np.random.default_rng(seed)
    # log-uniform magnitudes over the whole range where the double expression is inexact, denormals included
    e = rng.uniform(-149.0, -18.0, n_tiny)
    tiny = (np.exp2(e) * rng.choice([-1.0, 1.0], n_tiny)).astype(np.float32)
    unit = rng.uniform(-1.0, 1.0, n_unit).astype(np.float32)
    out = (rng.uniform(1.0, 4.0, n_out) * rng.choice([-1.0, 1.0], n_out)).astype(np.float32)
    return np.concatenate([tiny, unit, out])


def points(scalars, seed):
    """[N,3] points whose axes run through `scalars` in three independent orders (every scalar appears on every axis)."""
    rng = np.random.default_rng(seed)
    s = np.asarray(scalars, dtype=np.float32)
    return np.stack([s, s[rng.permutation(s.size)], s[rng.permutation(s.size)]], axis=1)
