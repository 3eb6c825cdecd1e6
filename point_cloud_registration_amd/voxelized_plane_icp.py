"""Voxelized point-to-plane ICP, the reference's ``VPlaneICP`` (``voxelized_plane_icp.py:12-64``)."""

from . import _capi
from .registration import Registration
from .voxel import VoxelGrid


class VPlaneICP(Registration):
    KIND = _capi.VPLANE

    def __init__(self, voxel_size=1.0, max_iter=30, max_dist=2, tol=1e-3, **kw):
        super().__init__(max_iter=max_iter, tol=tol, **kw)
        self.voxel_size = voxel_size
        self.max_dist = max_dist

    def set_target(self, target):
        self.voxels = VoxelGrid(self.voxel_size, device=self._device, _ctx=self._ctx())
        self.voxels.set_points(target)
        self._target = self.voxels._target
        self._is_target_set = True

    def calc_H_g_e2_no_parallel_ver(self, cur_T, source):
        """Per-point loop of the same sums (voxelized_plane_icp.py:67-101): nearest voxel centroid and
        voxel normal, gate on the centroid distance."""
        from .plane_icp import _plane_loop
        return _plane_loop(cur_T, source, self.voxels.kdtree.query, self.voxels.mean, self.voxels.norm, self.max_dist)
