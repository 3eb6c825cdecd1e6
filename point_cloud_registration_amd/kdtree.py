"""The reference's correspondence-search seam: ``KDTree(data).query(points, k=1) -> (dist, idx)``.

The reference picks a third-party CPU KD-tree by a module constant (``kdtree.py:6-65``:
pykdtree / small_gicp / scipy).  Here the one backend is the MI355X: an exact nearest-neighbour
search over a dense cell grid in HBM (csrc/nn_device.h).  Distances are Euclidean (not squared)
and ``idx`` indexes the array the tree was built on, as in every reference backend.
"""

import numpy as np

from . import _capi

USE_KDTREE = "MI355X_GRID"


class KDTree:
    def __init__(self, data, leafsize=16, device=None, _ctx=None):
        data = np.asarray(data)
        if data.ndim != 2 or data.shape[1] != 3:
            raise ValueError("data must have shape (N, 3)")
        self.n = data.shape[0]
        self._dtype = data.dtype if data.dtype.kind == "f" else np.dtype(np.float64)
        ctx = _ctx if _ctx is not None else _capi.get_context(device)
        # float32 data (the PCD case) is searched in float32; a float64 array is searched in float64, as every backend of
        # the reference does with the dtype it is given (kdtree.py:18-21): float32 filter search
        # over the index + float64 check / box search, the float64 search's neighbour in every case
        # Limits of the float64 emulation: QUERY points are taken as float32 (what every registration class passes,
        # registration.py:83) and up-cast; k > 1 searches the float32 copy of the data; and coordinates float32 cannot resolve
        # (UTM-scale clouds: a point moves by more than a quarter cell when rounded) keep the float32 search with a
        # RuntimeWarning instead of failing -- the reference's float64 tree would differ there only between near-equidistant
        # neighbours.
        self._target = _capi.Target.points(ctx, data.astype(np.float32, copy=False))
        if data.dtype == np.float64 and self.n > 0:
            self._target.set_points_f64(data)

    def query(self, points, k=1, distance_upper_bound=np.inf):
        """Exact k nearest neighbours: ``(dist, idx)``, shape (M,) for k=1 and (M, k) otherwise."""
        points = np.asarray(points)
        if k == 1:
            dist, idx = self._target.nn_query(points, distance_upper_bound)
        else:
            dist, idx = self._target.knn_query(points, k)
        return dist.astype(self._dtype, copy=False), idx
