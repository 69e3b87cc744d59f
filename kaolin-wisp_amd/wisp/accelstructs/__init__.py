from .base_as import BaseAS, ASQueryResults, ASRaytraceResults, ASRaymarchResults
from .octree_as import OctreeAS
from .aabb_as import AxisAlignedBBoxAS
