"""AxisAlignedBBoxAS: a single-cell bounding volume, i.e. a level-1 dense octree
(wisp/accelstructs/aabb_as.py:14-27)."""
import wisp.ops.spc as wisp_spc_ops
from wisp.accelstructs.octree_as import OctreeAS


class AxisAlignedBBoxAS(OctreeAS):
    def __init__(self):
        super().__init__(wisp_spc_ops.create_dense_octree(1))

    def name(self) -> str:
        return "AABB"
