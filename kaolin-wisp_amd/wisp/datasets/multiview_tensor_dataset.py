"""MultiviewTensorDataset: the on-device equivalent of NeRFSyntheticDataset's tensor layout - rays [V, H*W, 3] x 2 and
rgb [V, H*W, 3] resident in HBM (wisp/datasets/formats/nerf_standard_dataset.py:443-450) - with the same __getitem__
contract (one view -> MultiviewBatch, optionally passed through a transform such as SampleRays).  No image decoding or
camera model here: tensors are supplied by the caller (see synlego.py for the synthetic stand-in)."""
import torch

from wisp.core import Rays
from wisp.datasets.batch import MultiviewBatch


class MultiviewTensorDataset(torch.utils.data.Dataset):
    def __init__(self, origins, dirs, rgb, dist_min, dist_max, transform=None, img_shape=None):
        assert origins.shape == dirs.shape and origins.ndim == 3 and origins.shape[-1] == 3
        assert rgb is None or rgb.shape[:2] == origins.shape[:2]
        self.data = dict(rays=Rays(origins, dirs, dist_min=dist_min, dist_max=dist_max), rgb=rgb)
        self.transform = transform
        self.img_shape = img_shape

    def __len__(self):
        return self.data["rays"].origins.shape[0]

    def __getitem__(self, idx) -> MultiviewBatch:
        rays = self.data["rays"][idx]
        out = MultiviewBatch(rays=rays, rgb=None if self.data["rgb"] is None else self.data["rgb"][idx])
        if self.transform is not None:
            out = self.transform(out)
        return out
