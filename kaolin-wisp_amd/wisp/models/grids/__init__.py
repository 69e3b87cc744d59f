from .blas_grid import *
from .hash_grid import *
from .utils import MultiTable
