"""wisp - MI355X-native backend behind the kaolin-wisp plugin surface.

Import path, class names and constructor signatures follow NVIDIAGameWorks/kaolin-wisp for the volumetric
hot path (wisp.core, wisp.accelstructs, wisp.models.grids / nefs / decoders / embedders, wisp.tracers,
wisp.ops.grid, wisp.ops.spc); the compute is hand-written HIP for gfx950 in ../csrc behind the C ABI of
include/wisp_hip.h.  Viewer, config system, datasets and mesh tooling are intentionally not provided.
"""
__version__ = "0.1.0+mi355x"
