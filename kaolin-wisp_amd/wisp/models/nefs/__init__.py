from .base_nef import *
from .nerf import *
