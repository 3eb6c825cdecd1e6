"""Point-to-point ICP with the reference's ``ICP`` interface (``icp.py:12-57``)."""

import numpy as np

from . import _capi
from .kdtree import KDTree
from .math_tools import skew, transform_points
from .registration import Registration


class ICP(Registration):
    KIND = _capi.ICP

    def __init__(self, max_iter=30, max_dist=2, tol=1e-3, **kw):
        super().__init__(max_iter=max_iter, tol=tol, **kw)
        self.max_dist = max_dist

    def set_target(self, target):
        """float32 copy of the target + exact-NN index on the GPU (icp.py:17-22)."""
        target = np.asarray(target).astype(np.float32)
        self.kdtree = KDTree(target, device=self._device, _ctx=self._ctx())
        self.target = target
        self._target = self.kdtree._target        # the registration kernels share the tree's index
        self._is_target_set = True

    def calc_H_g_e2_no_parallel_ver(self, cur_T, source):
        """Per-point loop of the same sums, for reading and for tests (the reference keeps one too,
        icp.py:59-90).  Host Python over the GPU's correspondences; uses the consistent gradient
        J^T r, so it equals ``calc_H_g_e2`` at R = I and differs at R != I by quirk Q1."""
        cur_T = np.asarray(cur_T, dtype=np.float64)
        R = cur_T[:3, :3]
        src_trans = transform_points(cur_T.astype(np.float32), np.asarray(source, dtype=np.float32))
        dist, idx = self.kdtree.query(src_trans)
        H, g, e2 = np.zeros((6, 6)), np.zeros(6), 0.0
        for i in np.nonzero(dist < self.max_dist)[0]:
            J = np.hstack([np.eye(3), -R @ skew(np.asarray(source[i], dtype=np.float64))])
            r = (src_trans[i] - self.target[idx[i]]).astype(np.float64)
            H += J.T @ J
            g += J.T @ r
            e2 += r @ r
        return H, g, e2
