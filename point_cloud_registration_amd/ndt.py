"""Mahalanobis least-squares "NDT", the reference's ``NDT`` (``ndt.py:12-57``)."""

from . import _capi
from .registration import Registration
from .voxel import VoxelGrid


class NDT(Registration):
    KIND = _capi.NDT

    def __init__(self, voxel_size=1.0, max_iter=30, max_dist=2, tol=1e-3, **kw):
        super().__init__(max_iter=max_iter, tol=tol, **kw)
        self.voxel_size = voxel_size
        self.max_dist = max_dist

    def set_target(self, target):
        self.voxels = VoxelGrid(self.voxel_size, device=self._device, _ctx=self._ctx())
        self.voxels.set_points(target)
        self.voxels.calc_icov()       # ndt.py:21 (the GPU build always produces icov; kept for the interface)
        self._target = self.voxels._target
        self._is_target_set = True
