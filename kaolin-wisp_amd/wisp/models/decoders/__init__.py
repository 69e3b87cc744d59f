from .basic_decoders import *
