"""``VoxelGrid`` and friends with the reference's interface (``voxel.py:12-241``).

``set_points`` builds everything on the GPU (csrc/voxel_build.hip): integer voxel hash
(``get_keys``), grouping, per-voxel mean, two-pass sample covariance, the ``min_points`` filter,
smallest-eigenvector normals, closed-form inverse covariances and an exact nearest-CENTROID
index -- ``query`` returns the nearest kept voxel's statistics, not the voxel containing the
point, exactly like the reference's KD-tree over centroids (``voxel.py:165,171-179``).
"""

import numpy as np

from . import _capi

HASH_P = 116101
MAX_N = 10000000000


def get_keys(points, voxel_size=1.0):
    """Integer voxel hash (voxel.py:12-21); host utility mirroring what the GPU build computes."""
    ijk = np.floor(np.asarray(points) / voxel_size).astype(np.int64)
    x, y, z = ijk[:, 0], ijk[:, 1], ijk[:, 2]
    return (((z * HASH_P) % MAX_N + y) * HASH_P) % MAX_N + x


class VoxelGrid:
    def __init__(self, voxel_size, min_points=10, device=None, _ctx=None):
        self.voxel_size = voxel_size
        self.min_points = min_points
        self.kdtree = None
        self._device = device
        self._ctx = _ctx
        self._target = None

    def set_points(self, points):
        ctx = self._ctx if self._ctx is not None else _capi.get_context(self._device)
        self._target = _capi.Target.voxels(ctx, np.asarray(points), self.voxel_size, self.min_points)
        st = self._target.voxel_stats(("mean", "cov", "norm", "icov"))
        self.mean, self.cov, self.norm = st["mean"], st["cov"], st["norm"]
        self._icov = st["icov"]
        self.kdtree = _CentroidTree(self._target, self.mean)

    def calc_icov(self):
        """voxel.py:69-102; already computed by the GPU build, exposed under the reference's name."""
        self.icov = self._icov

    def calc_sqrt_icov(self):
        """voxel.py:61-67 (unused by the registration classes)."""
        self.calc_icov()
        self.sqrt_icov = np.transpose(np.linalg.cholesky(self.icov), axes=(0, 2, 1))

    def query(self, points, names):
        """Nearest kept voxel of each point -> {name: stats[idx]} plus 'dist' (voxel.py:171-179)."""
        dist, idx = self.kdtree.query(points)
        if "icov" in names and not hasattr(self, "icov"):
            self.calc_icov()
        out = {name: getattr(self, name)[idx] for name in names}
        out["dist"] = dist
        return out


class _CentroidTree:
    """KDTree(means) of the reference (voxel.py:165): float64 nearest-centroid search on the GPU."""

    def __init__(self, target, means=None):
        self._target = target
        self._means = means
        self._knn = None

    def query(self, points, k=1):
        """k = 1: the exact float64 nearest-centroid search of the registration kernels.  k > 1 (off the hot path; the
        reference's KDTree(means) answers any k, voxel.py:165): the GPU's exact k-NN over the float32-rounded centroids
        nominates k + 2, their float64 distances to the TRUE centroids are recomputed and the k smallest kept in (distance,
        index) order -- the float64 tree's answer unless centroids beyond the (k + 2)-th lie within float32 rounding (~1e-6 of
        the coordinates) of the k-th."""
        points = np.asarray(points)
        if k == 1:
            return self._target.nn_query(points)
        means = self._means if self._means is not None else self._target.voxel_stats(("mean",))["mean"]
        kk = min(int(k) + 2, means.shape[0])
        if kk < k:
            raise ValueError(f"k = {k} exceeds the number of voxels ({means.shape[0]})")
        if self._knn is None:
            self._knn = _capi.Target.points(self._target.ctx if not isinstance(self._target.ctx, _capi.Group) else self._target.ctx.member(0),
                                            means.astype(np.float32))
        _, cand = self._knn.knn_query(points.astype(np.float32), kk)
        d = np.linalg.norm(points.astype(np.float64)[:, None, :] - means[cand], axis=2)
        order = np.lexsort((cand, d), axis=1)[:, :k]
        rows = np.arange(points.shape[0])[:, None]
        return d[rows, order], cand[rows, order]


def voxel_filter(points, voxel_size, device=None):
    """Voxel-grid down-sampling: one centroid per occupied voxel, ascending key order
    (voxel.py:209-241).  Runs on the GPU through the voxel build (hash, radix sort, per-voxel
    float64 mean) with ``min_points=1``; the result is cast to float32 as in the reference."""
    t = _capi.Target.voxels(_capi.get_context(device), np.asarray(points), voxel_size, 1)
    try:
        return t.voxel_stats(("mean",))["mean"].astype(np.float32)
    finally:
        t.close()


def color_by_voxel(points, voxel_size):
    """Per-voxel pseudo-colours for visualisation (voxel.py:183-206)."""
    points = np.asarray(points)
    keys = get_keys(points, voxel_size)
    uniq, inv = np.unique(keys, return_inverse=True)
    rng = np.random.RandomState(42)
    colors = rng.randint(0, 256, size=(len(uniq), 3)).astype(np.uint32)[inv]
    rgb = (colors[:, 0] << 16) | (colors[:, 1] << 8) | colors[:, 2]
    return np.rec.fromarrays([points.astype(np.float32), rgb.astype(np.uint32)],
                             dtype=[("xyz", "<f4", (3,)), ("irgb", "<u4")])
