from .positional_embedder import *
