"""k-NN PCA normals with the reference's interface (``estimate_normals.py:11-110``)."""

import numpy as np

from .kdtree import KDTree


def estimate_normals(points, k=15, compat=True):
    """Normals (N,3) float32 from the k nearest neighbours of every point (GPU)."""
    tree = KDTree(points)
    return estimate_norm_with_tree(points, tree, k=k, compat=compat)


def estimate_norm_with_tree(points, kdtree, k=15, compat=True):
    """``compat=True`` keeps the reference's float32 single-pass covariance
    (estimate_normals.py:56-72); ``False`` uses a centred float64 covariance."""
    points = np.asarray(points)
    if not isinstance(kdtree, KDTree) or kdtree.n != points.shape[0]:
        kdtree = KDTree(points)
    return kdtree._target.estimate_normals(k, compat=compat)


def get_norm_lines(points, normals, length=0.1):
    """Line segments point -> point + length * normal for visualisation (estimate_normals.py:91-106)."""
    points = np.asarray(points)
    lines = np.empty((2 * points.shape[0], points.shape[1]), dtype=points.dtype)
    lines[0::2] = points
    lines[1::2] = points + np.asarray(normals) * length
    return lines
