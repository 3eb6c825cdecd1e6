"""Point-to-plane ICP with the reference's ``PlaneICP`` interface (``plane_icp.py:13-69``)."""

import numpy as np

from . import _capi
from .kdtree import KDTree
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

        ``kdree`` keeps the reference's (misspelt) keyword.  If both a tree and normals are
        given the normal estimation is skipped, as in the reference; a tree that is this
        package's :class:`KDTree` over the same cloud is reused, any other object is ignored
        and a GPU index is built (a foreign CPU tree cannot be searched from a HIP kernel).
        """
        target = np.asarray(target)
        self.target = target.astype(np.float32)
        if isinstance(kdree, KDTree) and kdree.n == self.target.shape[0]:
            self.kdtree = kdree
        else:
            self.kdtree = KDTree(self.target, device=self._device, _ctx=self._ctx())
        if kdree is None or norm is None:
            # k-NN PCA normals on the GPU (estimate_normals.py:27-87)
            self.normal = self.kdtree._target.estimate_normals(self.k, compat=self._compat_normals)
        else:
            self.normal = np.asarray(norm)
            self.kdtree._target.set_normals(self.normal)
        self._target = self.kdtree._target
        self._is_target_set = True
