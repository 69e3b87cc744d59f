"""SDFTensorDataset: (coordinate, signed distance) pairs resident in HBM with the item contract of the reference's SDF datasets
(wisp/datasets/formats/sdf datasets return `SDFBatch(coords=..., sdf=...)` per index, wisp/datasets/batch.py:75-110) plus
`get_batch(indices)`, which the trainer's loader uses to read a whole batch with one indexed load.  Sampling a mesh into such
pairs (OctreeSampledSDFDataset: mesh2sdf) is dataset preparation and out of scope; tensors come from the caller."""
import torch

from wisp.datasets.batch import SDFBatch


class SDFTensorDataset(torch.utils.data.Dataset):
    def __init__(self, coords, sdf, rgb=None, transform=None):
        assert coords.ndim == 2 and coords.shape[1] == 3 and sdf.shape[0] == coords.shape[0]
        self.data = dict(coords=coords, sdf=sdf.reshape(-1, 1), rgb=rgb)
        self.transform = transform
        self.sample_tex = rgb is not None

    @property
    def device(self):
        return self.data["coords"].device

    def __len__(self):
        return self.data["coords"].shape[0]

    def __getitem__(self, idx) -> SDFBatch:
        out = SDFBatch(coords=self.data["coords"][idx], sdf=self.data["sdf"][idx],
                       rgb=None if self.data["rgb"] is None else self.data["rgb"][idx])
        return self.transform(out) if self.transform is not None else out

    def get_batch(self, indices) -> SDFBatch:
        return self[indices]
