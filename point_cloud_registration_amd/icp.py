"""Point-to-point ICP with the reference's ``ICP`` interface (``icp.py:12-57``)."""

import numpy as np

from . import _capi
from .kdtree import KDTree
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
