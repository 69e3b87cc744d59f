"""TEST INFRASTRUCTURE ONLY - CPU restatement of wisp/ops/raygen/raygen.py:16-119 in numpy float32, one rounding per op
in the reference's order.  The camera transform is the part Kaolin owns (Camera.extrinsics.inv_transform_rays; source not
in /root/reference): it is restated as origin' = R^T (o - t), dir' = R^T d for the view matrix [R | t] (parity unpinned).
Parity: everything but that one leaf is PINNED to generate_centered_pixel_coords / generate_pinhole_rays / generate_ortho_rays compiled
from the reference file (tests/golden/raygen_ref.npz, 2e-6)."""
import numpy as np

f32 = np.float32


def centered_pixel_coords(img_width, img_height, res_x=None, res_y=None):
    res_x = img_width if res_x is None else res_x
    res_y = img_height if res_y is None else res_y
    py, px = np.meshgrid(np.arange(res_y, dtype=f32), np.arange(res_x, dtype=f32), indexing='ij')
    return (py * f32(float(img_height) / res_y) + f32(0.5)).astype(f32), (px * f32(float(img_width) / res_x) + f32(0.5)).astype(f32)


def generate_rays(pixel_x, pixel_y, ortho, x0, y0, width, height, sx, sy, view_rotation, view_translation):
    px, py = pixel_x.astype(f32).reshape(-1), pixel_y.astype(f32).reshape(-1)
    R = np.asarray(view_rotation, dtype=f32).reshape(3, 3)
    t = np.asarray(view_translation, dtype=f32).reshape(3)
    if not ortho:
        px = px - f32(x0)
        py = py + f32(y0)
    px = f32(2) * (px / f32(width)) - f32(1)
    py = f32(2) * (py / f32(height)) - f32(1)
    n = px.shape[0]
    o = np.zeros((n, 3), f32)
    d = np.zeros((n, 3), f32)
    if ortho:
        o[:, 0] = px * f32(sx)
        o[:, 1] = -(py * f32(sy))
        d[:, 2] = f32(-1)
    else:
        d[:, 0] = px * f32(sx)
        d[:, 1] = (-py) * f32(sy)
        d[:, 2] = f32(-1)
    q = o - t[None, :]
    ow = np.zeros_like(o)
    dw = np.zeros_like(d)
    for c in range(3):
        ow[:, c] = (R[0, c] * q[:, 0] + R[1, c] * q[:, 1]) + R[2, c] * q[:, 2]
        dw[:, c] = (R[0, c] * d[:, 0] + R[1, c] * d[:, 1]) + R[2, c] * d[:, 2]
    nrm = np.sqrt((dw[:, 0] * dw[:, 0] + dw[:, 1] * dw[:, 1]) + dw[:, 2] * dw[:, 2]).astype(f32)
    return ow, (dw / nrm[:, None]).astype(f32)
