from .conversions import *
from .constructors import *
from .processing import *
from .sampling import *
