from .blas_grid import *
from .octree_grid import *
from .codebook_grid import *
from .hash_grid import *
from .utils import MultiTable
from .triplanar_grid import *
