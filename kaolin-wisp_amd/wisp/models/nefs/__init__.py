from .base_nef import *
from .nerf import *
from .neural_sdf import *
