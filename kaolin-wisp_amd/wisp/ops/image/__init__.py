"""Image metrics of the validation loop (wisp/ops/image/metrics.py:19-37)."""
import numpy as np
import torch


def psnr(rgb, gts):
    """10 log10(1 / mse) for images in [0,1], shapes [..., 3]."""
    assert (rgb.max() <= 1.05 and rgb.min() >= -0.05)
    assert (gts.max() <= 1.05 and gts.min() >= -0.05)
    assert (rgb.shape[-1] == 3) and (gts.shape[-1] == 3)
    mse = torch.mean((rgb[..., :3] - gts[..., :3]) ** 2).item()
    return 10 * np.log10(1.0 / mse)
