"""OctreeAS: sparse-octree (SPC) occupancy structure; query / raytrace / raymarch run as HIP kernels.

Drop-in for wisp/accelstructs/octree_as.py:37-437: same constructors, attributes (octree, points, pyramid,
prefix, max_level, extent) and result layouts.  Differences that matter on MI355X:
  * 'ray' raymarch never materialises the R x N candidate tensors (octree_as.py:272-298): a count kernel keeps a
    1-bit-per-candidate mask, a scan turns per-ray counts into offsets, an emit kernel writes only survivors;
  * the occupancy test is a lookup in a one-bit-per-cell field of the marching level, built once per structure;
  * `raymarch(..., jitter=...)` optionally injects the stratification jitter so identical ray batches give
    identical samples (the reference is unseeded).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch

import wisp.ops.spc as wisp_spc_ops
from wisp.accelstructs.base_as import BaseAS, ASQueryResults, ASRaytraceResults, ASRaymarchResults


def _hip():
    import wisp._C as _C      # raises ImportError when libwisp_hip.so is not built: no fallback
    return _C


class OctreeAS(BaseAS):
    """Bottom-level acceleration structure over a Structured Point Cloud octree."""

    def __init__(self, octree):
        """octree (torch.ByteTensor): one occupancy byte per non-leaf node, breadth first, Morton order."""
        super().__init__()
        self.octree = octree
        parts = getattr(octree, '_wisp_spc_parts', None)     # set by the device build: hierarchy already derived
        if parts is not None:
            del octree._wisp_spc_parts               # (the derived tensors do not stay attached to the caller's tensor)
        if parts is not None and (len(parts) == 3 or parts[3] == octree._version):
            self.points, self.pyramid, self.prefix = parts[:3]      # an in-place edit since the build bumps _version: re-derive
        else:
            self.points, self.pyramid, self.prefix = wisp_spc_ops.octree_to_spc(octree)
        self.max_level = self.pyramid.shape[-1] - 2
        self.extent = dict()
        self._occ_bits = {}          # level -> occupancy bitfield (device tensor), built lazily

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_mesh(cls, mesh_path: str, level: int, sample_tex: bool = False,
                  num_samples_on_mesh: int = 100000000) -> OctreeAS:
        """Occupancy from samples over the faces of an OBJ mesh (octree_as.py:65-106): load, sphere-normalise, sample
        `num_samples_on_mesh` surface points (+ a half-cell jittered copy), quantise to `level`.  Sampling-based, hence not
        deterministic and not guaranteed hole-free - exactly the reference's caveat.  `sample_tex` (texture / material
        loading, documented as unused) is not provided."""
        from wisp.ops import mesh as mesh_ops
        if sample_tex:
            raise NotImplementedError("OctreeAS.from_mesh(sample_tex=True): textures / materials are not read by this backend")
        vertices, faces = mesh_ops.load_obj(mesh_path)
        vertices, faces = mesh_ops.normalize(vertices, faces, 'sphere')
        accel_struct = cls(wisp_spc_ops.mesh_to_octree(vertices, faces, level, num_samples_on_mesh))
        accel_struct.extent['vertices'] = vertices
        accel_struct.extent['faces'] = faces
        return accel_struct

    @classmethod
    def from_pointcloud(cls, pointcloud: torch.FloatTensor, level: int) -> OctreeAS:
        """Cells of `level` containing at least one point of `pointcloud` ([N,3] in [-1,1]) are occupied."""
        return cls(wisp_spc_ops.pointcloud_to_octree(pointcloud, level, dilate=0))

    @classmethod
    def from_quantized_points(cls, quantized_points: torch.LongTensor, level: int) -> OctreeAS:
        """quantized_points: integer cell coordinates [N,3] in [0, 2**level)."""
        built = wisp_spc_ops.build_spc(level, points=quantized_points)
        if built is not None:
            return cls._from_spc(*built)
        return cls(wisp_spc_ops.unbatched_points_to_octree(quantized_points, level, sorted=False))

    @classmethod
    def from_leaf_mask(cls, leaf_mask: torch.Tensor, level: int) -> OctreeAS:
        """Octree whose level-`level` cells are the non-zero entries of `leaf_mask` (u8 / bool [8^level], MORTON order:
        the order of the finest level of any point hierarchy).  The prune path: an occupancy test over the dense cells
        turns into the new structure without a sort or a gather.  Returns None when the mask is empty."""
        built = wisp_spc_ops.build_spc(level, leaf_mask=leaf_mask)
        return None if built is None else cls._from_spc(*built)

    @classmethod
    def _from_spc(cls, octree, points, pyramid, prefix) -> OctreeAS:
        octree._wisp_spc_parts = (points, pyramid, prefix, octree._version)
        return cls(octree)

    @classmethod
    def make_dense(cls, level: int) -> OctreeAS:
        """Fully occupied octree of depth `level`."""
        return cls(wisp_spc_ops.create_dense_octree(level))

    # ------------------------------------------------------------------ state helpers
    def _to_device(self, device):
        if self.octree.device != device:
            self.octree = self.octree.to(device)
            self.points = self.points.to(device)
            self.prefix = self.prefix.to(device)
            self._occ_bits = {}

    def _bitfield(self, level):
        bits = self._occ_bits.get(level)
        if bits is None and level <= 10:
            pts = wisp_spc_ops.unbatched_get_level_points(self.points, self.pyramid, level)
            bits = _hip().spc_bitfield(pts, level)
            self._occ_bits[level] = bits
        return bits

    def _coarse_bitfield(self, rays, num_samples, level):
        """(bits, level) of the coarser occupancy level the 'ray' count kernel stages in LDS to skip empty stretches of a
        ray, or (None, 0).  Pure work pruning: results are identical with and without it."""
        lc = _hip().raymarch_coarse_level(rays.dist_min, rays.dist_max, num_samples, level)
        if lc is None or self._bitfield(level) is None:
            return None, 0
        return self._bitfield(lc), lc

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_occ_bits'] = {}
        return state

    # ------------------------------------------------------------------ queries
    def query(self, coords, level=None, with_parents=False) -> ASQueryResults:
        """Cell index of `level` containing each coordinate ([N,3] in [-1,1]), -1 if empty / outside."""
        if level is None:
            level = self.max_level
        self._to_device(coords.device)
        return ASQueryResults(pidx=_hip().spc_query(self.octree, self.prefix, coords, level, with_parents))

    def query_chain(self, coords, level=None, first_level=0, hint=None, hint_group=1):
        """query(coords, level, with_parents=True).pidx[..., first_level:] without computing the columns in front of it: the
        cell of every level first_level .. level ([N, level - first_level + 1], -1 = empty / outside).  hint: see
        wisp_spc_query_chain (cells of first_level a march already knows; they never change the result)."""
        if level is None:
            level = self.max_level
        self._to_device(coords.device)
        return _hip().spc_query_chain(self.octree, self.prefix, self.points, coords, level, first_level, hint, hint_group)

    def raytrace_begin(self, rays, level=None):
        """The parameter-free, read-back-free first half of raytrace() (counts and offsets); hand the result to
        raytrace(..., begun=...).  A trainer issues it one batch ahead so that the size read-back never drains the GPU."""
        if level is None:
            level = self.max_level
        self._to_device(rays.origins.device)
        st = _hip().spc_raytrace_begin(self.octree, self.points, self.prefix, rays.origins, rays.dirs, level)
        st["blas"], st["rays"] = self, rays
        return st

    def _begun_fits(self, begun, rays, level):
        return (begun is not None and begun.get("blas") is self and begun.get("rays") is rays and begun.get("level") == level
                and "offsets" in begun)

    def raytrace(self, rays, level=None, with_exit=False, begun=None) -> ASRaytraceResults:
        """All ray / cell intersections at `level`, sorted by ray then front to back."""
        if level is None:
            level = self.max_level
        if not self._begun_fits(begun, rays, level):      # nothing (valid) was issued ahead for these rays on this octree
            begun = self.raytrace_begin(rays, level)
        ridx, pidx, depth, offsets = _hip().spc_raytrace_finish(begun, with_exit)
        res = ASRaytraceResults(ridx=ridx, pidx=pidx, depth=depth)
        res.ray_offsets = offsets     # nugget range of every ray; used by 'uniform' raymarch
        return res

    # ------------------------------------------------------------------ raymarch
    @staticmethod
    def _draw_seed():
        return int(torch.randint(0, 2 ** 62, (1,)).item())   # follows torch.manual_seed

    def _raymarch_voxel(self, rays, num_samples, level=None, jitter=None, begun=None) -> ASRaymarchResults:
        """num_samples jittered samples inside every intersected cell (S = nuggets * num_samples)."""
        rt = self.raytrace(rays, level, with_exit=True, begun=begun)
        ridx, samples, depth, deltas, boundary = _hip().raymarch_voxel(
            rays.origins, rays.dirs, rt.ridx, rt.depth, num_samples, jitter, self._draw_seed())
        res = ASRaymarchResults(ridx=ridx, samples=samples, depth_samples=depth, deltas=deltas, boundary=boundary,
                                pack_info=None)
        res.ray_offsets = rt.ray_offsets * num_samples        # every nugget contributes exactly num_samples samples
        # the cell every run of num_samples consecutive samples was generated in (a hint for spc_query_chain)
        res.nugget_pidx, res.nugget_level, res.samples_per_nugget = rt.pidx, (self.max_level if level is None else level), num_samples
        return res

    def _raymarch_ray(self, rays, num_samples, level=None, jitter=None) -> ASRaymarchResults:
        """num_samples stratified samples between dist_min and dist_max, keeping those inside occupied cells."""
        if torch.is_tensor(rays.dist_min) or torch.is_tensor(rays.dist_max):
            raise TypeError("'ray' raymarch needs scalar Rays.dist_min / dist_max (as the reference, octree_as.py:276-277)")
        self._to_device(rays.origins.device)
        coarse, lc = self._coarse_bitfield(rays, num_samples, level)
        ridx, samples, depth, deltas, boundary, offsets = _hip().raymarch_ray(
            self._bitfield(level), self.octree, self.prefix, rays.origins, rays.dirs, rays.dist_min, rays.dist_max,
            num_samples, level, jitter, self._draw_seed(), coarse, lc)
        res = ASRaymarchResults(ridx=ridx, samples=samples, depth_samples=depth, deltas=deltas, boundary=boundary,
                                pack_info=None)
        res.ray_offsets = offsets
        return res

    def _raymarch_uniform(self, rays, num_samples, level=None, begun=None) -> ASRaymarchResults:
        """Fixed world-space lattice of spacing ~2*sqrt(3)/num_samples clipped to the intersected cells."""
        rt = self.raytrace(rays, level, with_exit=True, begun=begun)
        step_size = 2 * np.sqrt(3) / num_samples
        scale = int(np.ceil(1.0 / step_size))
        step_size = 1.0 / float(scale)
        ridx, samples, depth, boundary, sample_offsets = _hip().raymarch_uniform(rays.origins, rays.dirs, rt.ridx, rt.depth,
                                                                                 rt.ray_offsets, scale)
        deltas = torch.full((ridx.shape[0], 1), step_size, dtype=torch.float32, device=depth.device)
        res = ASRaymarchResults(ridx=ridx, samples=samples, depth_samples=depth, deltas=deltas, boundary=boundary)
        res.ray_offsets = sample_offsets.index_select(0, rt.ray_offsets)    # per-nugget sample offsets at each ray's first nugget
        return res

    def raymarch(self, rays, raymarch_type, num_samples, level=None, jitter=None, begun=None, begin_only=False) -> ASRaymarchResults:
        """Generate packed samples along `rays`; raymarch_type in {'voxel', 'ray', 'uniform'}.
        begin_only: issue only the march's read-back-free prefix for `rays` (the cell intersection counts of 'voxel' and
        'uniform') and return its state - or None where the march has none; `begun`: such a state, issued earlier for the same
        Rays object (ignored when it does not fit: other rays, other level, an octree replaced by a prune)."""
        if level is None:
            level = self.max_level
        if begin_only:
            return self.raytrace_begin(rays, level) if raymarch_type in ('voxel', 'uniform') else None
        if raymarch_type == 'voxel':
            return self._raymarch_voxel(rays=rays, num_samples=num_samples, level=level, jitter=jitter, begun=begun)
        elif raymarch_type == 'ray':
            return self._raymarch_ray(rays=rays, num_samples=num_samples, level=level, jitter=jitter)
        elif raymarch_type == 'uniform':
            return self._raymarch_uniform(rays=rays, num_samples=num_samples, level=level, begun=begun)
        raise TypeError(f"Raymarch sampler type: {raymarch_type} is not supported by OctreeAS.")

    # ------------------------------------------------------------------ stats
    def occupancy(self) -> List[int]:
        return self.pyramid[0, :-2].cpu().numpy().tolist()

    def capacity(self) -> List[int]:
        return [8 ** lod for lod in range(self.max_level)]

    def name(self) -> str:
        return "Octree"
