"""Ray generation (wisp/ops/raygen/raygen.py:16-119): pixel grids and the pinhole / orthographic rays of one camera.

The per-pixel arithmetic is one HIP launch (csrc/render.hip, `wisp_generate_rays`) instead of ~15 tensor ops.  The
reference takes a kaolin `Camera`; Kaolin is not a dependency here, so the functions accept any object that offers what
they read from it - `width`, `height`, `x0`, `y0`, `near`, `far`, `tan_half_fov(axis)` (axis: 'horizontal' / 'vertical'
or kaolin's CameraFOV members), `fov_distance` (ortho) and the world->camera transform as `view_matrix()` ([1,4,4] or
[4,4]) - which a kaolin Camera does, and so does the small `LookAtCamera` below (what datasets and the validation
renderer of this package use).
"""
from dataclasses import dataclass
from typing import Sequence

import numpy as np
import torch

from wisp.core import Rays


def _hip():
    import wisp._C as _C
    return _C


# -- pixel grids (raygen.py:16-30) --
def generate_default_grid(width, height, device=None):
    h_coords = torch.arange(height, device=device, dtype=torch.float)
    w_coords = torch.arange(width, device=device, dtype=torch.float)
    return torch.meshgrid(h_coords, w_coords, indexing='ij')          # pixel_y, pixel_x


def generate_centered_pixel_coords(img_width, img_height, res_x=None, res_y=None, device=None):
    """(pixel_y, pixel_x) of a res_y x res_x grid covering an img_height x img_width image, at pixel centres."""
    res_x = img_width if res_x is None else res_x
    res_y = img_height if res_y is None else res_y
    pixel_y, pixel_x = generate_default_grid(res_x, res_y, device)
    pixel_x = pixel_x * (float(img_width) / res_x) + 0.5
    pixel_y = pixel_y * (float(img_height) / res_y) + 0.5
    return pixel_y, pixel_x


# -- camera access --
def _axis_tan(camera, horizontal: bool):
    try:
        return float(camera.tan_half_fov('horizontal' if horizontal else 'vertical'))
    except (TypeError, ValueError, KeyError):                                    # a kaolin Camera wants its enum
        from kaolin.render.camera.intrinsics import CameraFOV
        return float(camera.tan_half_fov(CameraFOV.HORIZONTAL if horizontal else CameraFOV.VERTICAL))


def _view_transform(camera):
    m = camera.view_matrix()
    m = torch.as_tensor(m, dtype=torch.float32).detach().cpu().reshape(-1, 4, 4)
    if m.shape[0] != 1:
        raise Exception("ray generation expects a single camera")
    m = m[0].numpy()
    return m[:3, :3].reshape(-1), m[:3, 3]


def _scalar(v):
    return float(v.reshape(-1)[0]) if torch.is_tensor(v) else float(v)


def _generate(camera, coords_grid, ortho):
    pixel_y, pixel_x = coords_grid
    if pixel_x.device != pixel_y.device:
        raise Exception(f"Expected coords_grid[0] and coords_grid[1] on the same device, but found {pixel_y.device} and {pixel_x.device}.")
    cam_dev = getattr(camera, 'device', None)
    if cam_dev is not None and torch.device(cam_dev) != pixel_x.device:
        raise Exception(f"Expected camera and coords_grid[0] to be on the same device, but found {cam_dev} and {pixel_x.device}.")
    rot, trans = _view_transform(camera)
    if ortho:
        aspect = _scalar(camera.width) / _scalar(camera.height)
        dist = _scalar(camera.fov_distance)
        sx, sy, x0, y0 = np.float32(dist) * np.float32(aspect), dist, 0.0, 0.0
    else:
        sx, sy = _axis_tan(camera, True), _axis_tan(camera, False)
        x0, y0 = _scalar(camera.x0), _scalar(camera.y0)
    origins, dirs = _hip().generate_rays(pixel_x, pixel_y, ortho, x0, y0, _scalar(camera.width), _scalar(camera.height),
                                         sx, sy, rot, trans)
    return Rays(origins=origins, dirs=dirs, dist_min=_scalar(camera.near), dist_max=_scalar(camera.far))


def generate_pinhole_rays(camera, coords_grid):
    """Rays through `coords_grid` = (pixel_y, pixel_x) for a pinhole camera whose principal point is displaced by
    (camera.x0, camera.y0) pixels from the image centre.  Returns wisp.core.Rays with [H*W, 3] origins / unit dirs."""
    return _generate(camera, coords_grid, ortho=False)


def generate_ortho_rays(camera, coords_grid):
    """Parallel rays (direction = camera -z) starting on the image plane scaled by camera.fov_distance."""
    return _generate(camera, coords_grid, ortho=True)


@dataclass
class LookAtCamera:
    """Minimal single camera with the attribute surface the ray generators read (the part of kaolin's Camera that
    Camera.from_args(eye, at, up, fov, width, height, near, far) fills): looks down its -z axis, y up."""
    eye: Sequence[float]
    at: Sequence[float]
    up: Sequence[float]
    fov: float                      # horizontal field of view, radians
    width: int
    height: int
    near: float = 1e-2
    far: float = 1e2
    x0: float = 0.0
    y0: float = 0.0
    fov_distance: float = 1.0       # orthographic mode only
    device: str = None

    def view_matrix(self):
        eye, at, up = (np.asarray(v, dtype=np.float64) for v in (self.eye, self.at, self.up))
        back = eye - at
        back /= np.linalg.norm(back)
        right = np.cross(up, back)
        right /= np.linalg.norm(right)
        true_up = np.cross(back, right)
        m = np.eye(4)
        m[0, :3], m[1, :3], m[2, :3] = right, true_up, back
        m[:3, 3] = -m[:3, :3] @ eye
        return torch.from_numpy(m.astype(np.float32))[None]

    def tan_half_fov(self, axis='horizontal'):
        t = float(np.tan(np.float64(self.fov) / 2.0))
        return t if str(axis).lower().endswith('horizontal') else t * (float(self.height) / float(self.width))
