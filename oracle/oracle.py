"""ctypes front-end of the CPU oracle (oracle/pcr_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg -- never from ``point_cloud_registration_amd``.
Parity status: pinned against golden vectors generated from the reference
(tests/golden/make_golden.py, tests/test_oracle_golden.py).

The composition functions at the bottom (``Target*``, ``calc_H_g_e2``, ``align``) restate
the reference's class-level flow: ``set_target`` (icp.py:17-22, plane_icp.py:19-28,
voxelized_plane_icp.py:18-21, ndt.py:18-22) and ``Registration.align``
(registration.py:71-113).
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ICP, PLANE, VPLANE, NDT = 0, 1, 2, 3
FLAG_ICP_RR_QUIRK = 1
FLAG_GATE_F64 = 2       # quirk Q6: float64 gate on the float64 tree's distances (plane_icp.py:22,41)

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build():
    """Compile the oracle with gcc (idempotent)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("PCR_ORACLE_LIB") or os.path.join(_HERE, "libpcr_oracle.so")   # (sanitizer build in tests)
    src = os.path.join(_HERE, "pcr_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build()
    L = C.CDLL(path)
    L.orc_max_threads.restype = C.c_int
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_transform.argtypes = [_f64p, _f32p, C.c_int64, _f32p]
    L.orc_nn_brute_f32.argtypes = [_f32p, C.c_int64, _f32p, C.c_int64, _f32p, _i64p]
    L.orc_nn_brute_f64.argtypes = [_f64p, C.c_int64, _f32p, C.c_int64, _f64p, _i64p]
    L.orc_knn_brute_f32.argtypes = [_f32p, C.c_int64, _f32p, C.c_int64, C.c_int, _f32p, _i64p]
    L.orc_grid_build.restype = C.c_void_p
    L.orc_grid_build.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_double]
    L.orc_grid_free.argtypes = [C.c_void_p]
    L.orc_grid_nn.argtypes = [C.c_void_p, _f32p, C.c_int64, C.c_double, _f64p, _i64p]
    L.orc_linearize.restype = C.c_int
    L.orc_linearize.argtypes = [C.c_int, _f64p, _f32p, _f32p, C.c_int64, C.c_void_p, C.c_void_p,
                                _f64p, _i64p, C.c_double, C.c_uint, _f64p]
    L.orc_voxel_keys.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_double, _i64p]
    L.orc_voxel_build.restype = C.c_int
    L.orc_voxel_build.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_int,
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_eigh3.argtypes = [_f64p, _f64p, _f64p]
    L.orc_calc_icov.argtypes = [_f64p, C.c_int64, _f64p]
    L.orc_normals_from_knn.argtypes = [_f32p, C.c_int64, _i64p, C.c_int, C.c_int, _f32p]
    L.orc_solve6.restype = C.c_int
    L.orc_solve6.argtypes = [_f64p, _f64p, _f64p]
    L.orc_expSO3.argtypes = [_f64p, _f64p]
    L.orc_plus.argtypes = [_f64p, _f64p, _f64p]
    _LIB = L
    return L


def max_threads():
    return int(lib().orc_max_threads())


def set_threads(n):
    """OpenMP team size of every later call (bench.py sets it to the container's CPU quota)."""
    lib().orc_set_threads(int(n))


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ----------------------------------------------------------------------------- pieces
def transform(T, src):
    src = _c(src, np.float32)
    out = np.empty_like(src)
    lib().orc_transform(_c(T, np.float64).reshape(16), src, src.shape[0], out)
    return out


def nn_brute(tgt, q):
    tgt, q = _c(tgt, np.float32), _c(q, np.float32)
    d = np.empty(q.shape[0], np.float32)
    i = np.empty(q.shape[0], np.int64)
    lib().orc_nn_brute_f32(tgt, tgt.shape[0], q, q.shape[0], d, i)
    return d, i


def nn_brute_f64(tgt, q):
    tgt, q = _c(tgt, np.float64), _c(q, np.float32)
    d = np.empty(q.shape[0], np.float64)
    i = np.empty(q.shape[0], np.int64)
    lib().orc_nn_brute_f64(tgt, tgt.shape[0], q, q.shape[0], d, i)
    return d, i


def knn_brute(tgt, q, k):
    tgt, q = _c(tgt, np.float32), _c(q, np.float32)
    d = np.empty((q.shape[0], k), np.float32)
    i = np.empty((q.shape[0], k), np.int64)
    lib().orc_knn_brute_f32(tgt, tgt.shape[0], q, q.shape[0], k, d, i)
    return d, i


class Grid:
    """Exact 1-NN over a dense cell grid (same answers as nn_brute*, much faster)."""

    def __init__(self, pts, cell):
        self.is_f64 = np.asarray(pts).dtype == np.float64
        self.pts = _c(pts, np.float64 if self.is_f64 else np.float32)
        self._h = lib().orc_grid_build(self.pts.ctypes.data, int(self.is_f64), self.pts.shape[0], float(cell))

    def query(self, q, r_max=np.inf):
        q = _c(q, np.float32)
        d = np.empty(q.shape[0], np.float64)
        i = np.empty(q.shape[0], np.int64)
        lib().orc_grid_nn(self._h, q, q.shape[0], float(r_max), d, i)
        return d, i

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_grid_free(self._h)
            self._h = None


def linearize(kind, T, src, src_trans, rec_a, rec_b, dist, idx, max_dist, flags=FLAG_ICP_RR_QUIRK):
    """Reduce step given correspondences -> (H 6x6, g 6, e2, count)."""
    src, st = _c(src, np.float32), _c(src_trans, np.float32)
    rdt = np.float32 if kind in (ICP, PLANE) else np.float64
    rec_a = _c(rec_a, rdt)
    rec_b = _c(rec_b, rdt) if rec_b is not None else None
    out = np.zeros(29)
    rc = lib().orc_linearize(kind, _c(T, np.float64).reshape(16), src, st, src.shape[0],
                             rec_a.ctypes.data, rec_b.ctypes.data if rec_b is not None else None,
                             _c(dist, np.float64), _c(idx, np.int64), float(max_dist), int(flags), out)
    assert rc == 0
    return unpack29(out)


def unpack29(out):
    H = np.zeros((6, 6))
    iu = np.triu_indices(6)
    H[iu] = out[:21]
    H = H + np.triu(H, 1).T
    return H, out[21:27].copy(), float(out[27]), int(round(out[28]))


def voxel_keys(points, voxel_size):
    is64 = np.asarray(points).dtype == np.float64
    p = _c(points, np.float64 if is64 else np.float32)
    keys = np.empty(p.shape[0], np.int64)
    lib().orc_voxel_keys(p.ctypes.data, int(is64), p.shape[0], float(voxel_size), keys)
    return keys


def calc_icov(cov):
    cov = _c(cov, np.float64)
    out = np.empty_like(cov)
    lib().orc_calc_icov(cov.reshape(-1), cov.shape[0], out.reshape(-1))
    return out


def voxel_build(points, voxel_size, min_points=10):
    """VoxelGrid.set_points restatement -> dict(mean, cov, norm, counts, keys, n_unique)."""
    is64 = np.asarray(points).dtype == np.float64
    p = _c(points, np.float64 if is64 else np.float32)
    nu, nk = C.c_int64(0), C.c_int64(0)
    L = lib()
    L.orc_voxel_build(p.ctypes.data, int(is64), p.shape[0], float(voxel_size), int(min_points),
                      C.byref(nu), C.byref(nk), None, None, None, None, None)
    n = nk.value
    mean = np.empty((n, 3)); cov = np.empty((n, 3, 3)); norm = np.empty((n, 3))
    counts = np.empty(n, np.int64); keys = np.empty(n, np.int64)
    L.orc_voxel_build(p.ctypes.data, int(is64), p.shape[0], float(voxel_size), int(min_points),
                      C.byref(nu), C.byref(nk), mean.ctypes.data, cov.ctypes.data, norm.ctypes.data,
                      counts.ctypes.data, keys.ctypes.data)
    return {"mean": mean, "cov": cov, "norm": norm, "counts": counts, "keys": keys, "n_unique": nu.value}


def eigh3(A):
    ev = np.empty(3); evec = np.empty(9)
    lib().orc_eigh3(_c(A, np.float64).reshape(9), ev, evec)
    return ev, evec.reshape(3, 3)          # rows = eigenvectors, ascending eigenvalue


def normals_from_knn(points, knn_idx, compat=True):
    """One normal per ROW of knn_idx (indices into points): rows may be a subset of the cloud."""
    p = _c(points, np.float32)
    idx = _c(knn_idx, np.int64)
    out = np.empty((idx.shape[0], 3), np.float32)
    lib().orc_normals_from_knn(p, idx.shape[0], idx, idx.shape[1], int(bool(compat)), out)
    return out


def solve6(H, g):
    x = np.empty(6)
    if lib().orc_solve6(_c(H, np.float64).reshape(36), _c(g, np.float64), x):
        raise np.linalg.LinAlgError("Singular matrix")
    return x


def expSO3(w):
    R = np.empty(9)
    lib().orc_expSO3(_c(w, np.float64), R)
    return R.reshape(3, 3)


def plus(T, dx):
    out = np.empty(16)
    lib().orc_plus(_c(T, np.float64).reshape(16), _c(dx, np.float64), out)
    return out.reshape(4, 4)


# ----------------------------------------------------------------------------- composition
class TargetPoints:
    """ICP / PlaneICP target: float32 copy of the cloud (+ normals) and an exact NN index."""

    def __init__(self, target, normals=None, k=15, cell=None, compat_normals=True, tree_f64=False):
        """tree_f64 (quirk Q6, plane_icp.py:20-22): PlaneICP builds its KD-tree on the ORIGINAL array, so a float64
        target is SEARCHED in float64 (queries up-cast, float64 distances, float64 gate) while the matched records are
        gathered from the float32 copy (plane_icp.py:20,44).  Only meaningful when `target` is float64."""
        self.pts = _c(target, np.float32)
        self.tree_f64 = bool(tree_f64) and np.asarray(target).dtype == np.float64
        self.pts64 = _c(target, np.float64) if self.tree_f64 else None
        self._brute = self.pts.shape[0] <= 4096
        if not self._brute:
            if cell is None:
                ext = self.pts.max(0) - self.pts.min(0)
                cell = max(float(np.cbrt(np.prod(np.maximum(ext, 1e-3)) / max(self.pts.shape[0], 1)) * 2.0), 1e-3)
            self.grid = Grid(self.pts64 if self.tree_f64 else self.pts, cell)
        self.normals = None
        if normals is not None:
            self.normals = _c(normals, np.float32)

    def estimate_normals(self, k=15, compat=True):
        _, idx = knn_brute(self.pts, self.pts, k)
        self.normals = normals_from_knn(self.pts, idx, compat)
        return self.normals

    def query(self, q, r_max=np.inf):
        if self._brute:
            if self.tree_f64:
                return nn_brute_f64(self.pts64, q)
            d, i = nn_brute(self.pts, q)
            return d.astype(np.float64), i
        return self.grid.query(q, r_max)


class TargetVoxels:
    """VPlaneICP / NDT target: voxel statistics + exact nearest-centroid index (float64)."""

    def __init__(self, target, voxel_size=1.0, min_points=10):
        v = voxel_build(target, voxel_size, min_points)
        self.mean, self.cov, self.norm = v["mean"], v["cov"], v["norm"]
        self.icov = calc_icov(self.cov)
        self.icov6 = np.ascontiguousarray(self.icov.reshape(-1, 9)[:, [0, 1, 2, 4, 5, 8]])
        self._brute = self.mean.shape[0] <= 4096
        if not self._brute:
            self.grid = Grid(self.mean, voxel_size)

    def query(self, q, r_max=np.inf):
        if self._brute:
            return nn_brute_f64(self.mean, q)
        return self.grid.query(q, r_max)


def calc_H_g_e2(kind, target, cur_T, source, max_dist=2.0, flags=FLAG_ICP_RR_QUIRK, with_count=False):
    """One pass of the hot path (calc_H_g_e2 of the four reference classes)."""
    source = _c(source, np.float32)
    st = transform(cur_T, source)
    # bounded search is exact for the gated sums: anything at >= max_dist is masked anyway
    dist, idx = target.query(st, r_max=float(max_dist) * 1.0000001 + 1e-12)
    if kind == ICP:
        H, g, e2, cnt = linearize(kind, cur_T, source, st, target.pts, None, dist, idx, max_dist, flags)
    elif kind == PLANE:
        if getattr(target, "tree_f64", False):
            flags |= FLAG_GATE_F64
        H, g, e2, cnt = linearize(kind, cur_T, source, st, target.pts, target.normals, dist, idx, max_dist, flags)
    elif kind == VPLANE:
        H, g, e2, cnt = linearize(kind, cur_T, source, st, target.mean, target.norm, dist, idx, max_dist, flags)
    else:
        H, g, e2, cnt = linearize(kind, cur_T, source, st, target.mean, target.icov6, dist, idx, max_dist, flags)
    return (H, g, e2, cnt) if with_count else (H, g, e2)


def align(kind, target, source, init_T=None, max_iter=30, tol=1e-3, max_dist=2.0,
          flags=FLAG_ICP_RR_QUIRK, trace=None):
    """Gauss-Newton driver, registration.py:71-113 (quirks Q3, Q4, Q7 kept)."""
    cur_T = np.eye(4) if init_T is None else np.array(init_T, dtype=np.float64)
    source = _c(source, np.float32)
    for _ in range(max_iter):
        H, g, e2 = calc_H_g_e2(kind, target, cur_T, source, max_dist, flags)
        if trace is not None:
            trace.append((cur_T.copy(), H, g, e2))
        dx = -solve6(H, g)
        if np.linalg.norm(dx) < tol:          # convergence test precedes the update (Q4)
            break
        cur_T = plus(cur_T, dx)
    return cur_T
