"""MI355X-native drop-in for the registration hot path of scomup/point-cloud-registration.

Same public names as the reference's ``point_cloud_registration/__init__.py:1-10`` (minus the
experimental Caratheodory coreset helpers, which are not on any ``align()`` path): the O(N) work
of ``set_target`` / ``calc_H_g_e2`` / ``align`` runs in hand-written HIP kernels for gfx950 behind
a C ABI (include/pcr.h); there is no CPU fallback.
"""

from .registration import Registration
from .math_tools import (makeRt, expSO3, makeT, skews, huber_weight, plus, transform_points,
                         skew_time_vector, skew, skew2)
from .voxelized_plane_icp import VPlaneICP
from .plane_icp import PlaneICP
from .icp import ICP
from .ndt import NDT
from .kdtree import KDTree
from .voxel import VoxelGrid, voxel_filter, color_by_voxel, get_keys
from .estimate_normals import estimate_normals, get_norm_lines, estimate_norm_with_tree

__all__ = [
    "Registration", "ICP", "PlaneICP", "VPlaneICP", "NDT", "KDTree", "VoxelGrid", "voxel_filter",
    "color_by_voxel", "get_keys", "estimate_normals", "get_norm_lines", "estimate_norm_with_tree",
    "makeRt", "expSO3", "makeT", "skews", "huber_weight", "plus", "transform_points",
    "skew_time_vector", "skew", "skew2",
]
