from .base_nef import *
from .nerf import *
from .neural_sdf import *
from .image_nef import *
