"""Point-to-plane ICP with the reference's ``PlaneICP`` interface (``plane_icp.py:13-69``)."""

import numpy as np

from . import _capi
from .kdtree import KDTree
from .math_tools import skew, transform_points
from .registration import Registration


class PlaneICP(Registration):
    KIND = _capi.PLANE

    def __init__(self, max_iter=30, max_dist=2, tol=1e-3, k=15, compat_normals=True, **kw):
        super().__init__(max_iter=max_iter, tol=tol, **kw)
        self.max_dist = max_dist
        self.k = k
        self._compat_normals = compat_normals

    def set_target(self, target, kdree=None, norm=None):
        """Target + per-point normals (plane_icp.py:19-28).

        ``kdree`` keeps the reference's (misspelt) keyword.  If both a tree and normals are given the
        normal estimation is skipped, as in the reference.  The tree object itself is never searched or
        modified: a foreign CPU tree cannot be searched from a HIP kernel, and this registration's
        normals live in its own device index (built in a few milliseconds), so a tree the caller shares
        with other registrations keeps its state.
        """
        target = np.asarray(target)
        self.target = target.astype(np.float32)
        # quirk Q6 (plane_icp.py:22): the tree is built on the ORIGINAL array -- a float64 target is searched in float64
        # (KDTree keeps the float64 coordinates beside the float32 index), the records come from the float32 copy (:20,44)
        self.kdtree = KDTree(target if target.dtype == np.float64 else self.target, device=self._device, _ctx=self._ctx())
        if kdree is None or norm is None:
            # k-NN PCA normals on the GPU (estimate_normals.py:27-87).  They stay there: ``self.normal`` (the attribute the
            # reference sets, plane_icp.py:23-24) reads them back the first time somebody asks -- 12.7 MB over PCIe per
            # 1.06 M points, 0.35 ms of a 3.4 ms set_target that align() never needs
            self.kdtree._target.estimate_normals(self.k, compat=self._compat_normals, want=False)
            self._normal = None
        else:
            self._normal = np.asarray(norm)
            if self._normal.shape != self.target.shape:
                raise ValueError("norm must have the shape of the target")
            self.kdtree._target.set_normals(self._normal)
        self._target = self.kdtree._target
        self._is_target_set = True

    @property
    def normal(self):
        if getattr(self, "_normal", None) is None and getattr(self, "_target", None) is not None:
            self._normal = self._target.get_normals()
        return getattr(self, "_normal", None)

    @normal.setter
    def normal(self, value):
        self._normal = value

    def calc_H_g_e2_no_parallel_ver(self, cur_T, source):
        """Per-point loop of the same sums (the reference keeps one, plane_icp.py:72-101); host Python
        over the GPU's correspondences, gate on the point distance as in the vectorised path."""
        return _plane_loop(cur_T, source, self.kdtree.query, self.target, self.normal, self.max_dist)


def _plane_loop(cur_T, source, query, means, norms, max_dist):
    cur_T = np.asarray(cur_T, dtype=np.float64)
    R = cur_T[:3, :3]
    src_trans = transform_points(cur_T.astype(np.float32), np.asarray(source, dtype=np.float32))
    dist, idx = query(src_trans)
    H, g, e2 = np.zeros((6, 6)), np.zeros(6), 0.0
    for i in np.nonzero(dist < max_dist)[0]:
        n = np.asarray(norms[idx[i]], dtype=np.float64)
        r = float(n @ (src_trans[i] - means[idx[i]]))
        J = np.concatenate([n, skew(np.asarray(source[i], dtype=np.float64)) @ (R.T @ n)])
        H += np.outer(J, J)
        g += J * r
        e2 += r * r
    return H, g, e2
