from .batch import Batch, MultiviewBatch, SDFBatch
from .transforms import SampleRays
from .multiview_tensor_dataset import MultiviewTensorDataset
from .sdf_tensor_dataset import SDFTensorDataset
