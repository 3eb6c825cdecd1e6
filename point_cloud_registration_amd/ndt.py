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

    def calc_H_g_e2_no_parallel_ver(self, cur_T, source):
        """Per-point loop of the same sums (ndt.py:60-98)."""
        import numpy as np
        from .math_tools import skew, transform_points
        cur_T = np.asarray(cur_T, dtype=np.float64)
        R = cur_T[:3, :3]
        src_trans = transform_points(cur_T.astype(np.float32), np.asarray(source, dtype=np.float32))
        q = self.voxels.query(src_trans, ["icov", "mean"])
        H, g, e2 = np.zeros((6, 6)), np.zeros(6), 0.0
        for i in np.nonzero(q["dist"] < self.max_dist)[0]:
            J = np.hstack([np.eye(3), -R @ skew(np.asarray(source[i], dtype=np.float64))])
            r = src_trans[i].astype(np.float64) - q["mean"][i]
            C = q["icov"][i]
            H += J.T @ C @ J
            g += J.T @ C @ r
            e2 += r @ C @ r
        return H, g, e2
