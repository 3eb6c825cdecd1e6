"""Gauss-Newton driver with the reference's ``Registration`` interface.

Same names, arguments and error behaviour as ``point_cloud_registration/registration.py:10-113``:
``set_target`` / ``is_target_set`` / ``align(source, init_T, verbose)`` /
``calc_H_g_e2(cur_T, source) -> (H 6x6, g 6, e2)``.  The O(N) body of ``calc_H_g_e2`` is one
call into libpcr_hip.so (transform + exact NN + gate + residual/Jacobian + 6x6 reduction on the
MI355X); the 6x6 solve, the boxplus and the tolerance test stay on the host, as in the
reference.

Additions (none changes a default):
* ``device=`` -- which GPU this process drives (default ``LOCAL_RANK``);
* ``comm=`` -- a :class:`distributed.Communicator`; the scan passed to ``align`` is then this
  rank's SHARD and every ``calc_H_g_e2`` returns the sum over all ranks;
* ``devices=`` -- a list of GPU ids for ONE process (``PlaneICP(devices=[0, 1, 2, 3, 4, 5, 6, 7])``): ``set_target`` builds
  the target on every listed GPU, ``align`` / ``calc_H_g_e2`` take the WHOLE scan, shard it across them behind the C ABI
  (``pcr_group_*``) and exchange the 29 sums GPU to GPU -- the reference's single-process call order
  (registration.py:28,71; demo_matching.py:147-152) on a whole node, no torchrun (SURVEY.md sections 5 and 8b);
* ``upload(source)`` -- returns a handle to the device copy of a scan; ``calc_H_g_e2(cur_T, handle)`` and
  ``align(handle)`` then skip the upload and the per-call content hash of the array form;
* ``native_loop`` (default True) -- ``align`` runs the whole loop behind the C ABI (``pcr_align``):
  pose in HBM, solve + boxplus in a one-wave kernel behind the reduce kernel, iterations enqueued back to
  back, one result read by the host.  ``native_loop=False`` keeps the Python loop of the reference
  (one ``calc_H_g_e2`` + ``numpy.linalg.solve`` per iteration), which ``verbose=True`` also uses
  (it prints the reference's line before every solve).
"""

import numpy as np

from . import _capi
from .math_tools import plus


class UploadedScan:
    """A scan that already lives on the GPU (``Registration.upload``): pass it wherever ``source`` is expected
    to skip the per-call content hash of the array form."""

    def __init__(self, scan, shape):
        self._scan = scan
        self.shape = shape

    def close(self):
        self._scan.close()


class Registration:
    KIND = None           # _capi.ICP / PLANE / VPLANE / NDT in the subclasses

    def __init__(self, max_iter=30, tol=1e-3, device=None, comm=None, native_loop=True,
                 compat_flags=_capi.FLAG_ICP_RR_QUIRK, devices=None):
        self.max_iter = max_iter
        self.tol = tol
        self._is_target_set = False
        self._device = device
        self._comm = comm
        if devices is not None and (comm is not None or device is not None):
            raise ValueError("devices= (one process, several GPUs) excludes device= and comm= (one process per GPU)")
        self._group = _capi.get_group(devices) if devices is not None else None
        self._native_loop = native_loop
        self._flags = compat_flags
        self._target = None            # _capi.Target
        self._scan = None              # (_capi.Scan, key) cache for calc_H_g_e2(cur_T, source)
        self._scan_key = None
        self.last_iterations = 0
        self.last_correspondences = 0

    # -- reference interface -------------------------------------------------------------------
    def is_target_set(self):
        return self._is_target_set

    def set_target(self, target):
        self._is_target_set = True
        raise NotImplementedError("set_target is not implemented.")

    def update_target(self, target):
        # registration.py:36-43: an unimplemented stub in the reference as well
        raise NotImplementedError("update_target is not implemented.")

    def linearize(self, cur_T, source):
        # registration.py:45-53: no subclass of the reference implements it either (dead path)
        raise NotImplementedError("linearize is not implemented.")

    def calc_H_g_e2(self, cur_T, source):
        """Hessian (6x6), gradient (6) and squared error at ``cur_T`` for ``source`` (N,3)."""
        if not self._is_target_set:
            raise ValueError("Target is not set.")
        scan = self._scan_for(source)
        return self._linearize(np.asarray(cur_T, dtype=np.float64), scan)

    def upload(self, source):
        """Upload (and Morton-sort) ``source`` once; the returned handle can stand in for the array in
        ``calc_H_g_e2`` / ``align`` (the caller then owns the "has it changed?" question)."""
        src = np.asarray(source)
        if src.ndim != 2 or src.shape[1] != 3:
            raise ValueError("source must have shape (N, 3)")
        return UploadedScan(_capi.Scan(self._ctx(), src.astype(np.float32, copy=False)), src.shape)

    def align(self, source, init_T=np.eye(4), verbose=False):
        """Gauss-Newton alignment of ``source`` onto the target; returns the 4x4 float64 pose."""
        if self.is_target_set() is False:
            raise ValueError("Target is not set.")
        scan = self._scan_for(source, fresh=True)     # the reference copies the scan per call
        cur_T = np.array(init_T, dtype=np.float64)
        if self._native_loop and not verbose and not self._needs_host_reduce():
            T, iters, trace = _capi.align(self._target, scan, self.KIND, cur_T, self.max_iter, self.tol,
                                          self._max_dist(), self._call_flags(), want_trace=True)
            self.last_iterations = iters
            if iters:
                self.last_correspondences = int(round(trace[iters - 1, 16 + 28]))
            return T
        it = 0
        for it in range(self.max_iter):
            H, g, e2 = self._linearize(cur_T, scan)
            if verbose:
                print(f"iter {it}, error {e2}")
            dx = -np.linalg.solve(H, g)          # LinAlgError when H is singular (quirk Q7)
            if np.linalg.norm(dx) < self.tol:    # the test precedes the update (quirk Q4)
                break
            cur_T = plus(cur_T, dx)
        self.last_iterations = it + 1 if self.max_iter > 0 else 0
        return cur_T

    # -- internals -----------------------------------------------------------------------------
    def _ctx(self):
        if self._group is not None:
            return self._group
        if self._comm is not None and getattr(self._comm, "ctx", None) is not None:
            return self._comm.ctx
        return _capi.get_context(self._device)

    def _max_dist(self):
        return float(getattr(self, "max_dist", 2.0))

    def _needs_host_reduce(self):
        return self._comm is not None and not self._comm.in_library

    def _call_flags(self):
        """The collective is a per-call decision: only a Registration that was given ``comm=`` joins the
        all-reduce, whatever else shares the (process-wide) context."""
        if self._group is not None or (self._comm is not None and self._comm.in_library):
            return self._flags
        return self._flags | _capi.FLAG_LOCAL_ONLY

    @staticmethod
    def _digest(src):
        """64-bit hash of the WHOLE scan buffer (``pcr_hash64`` inside libpcr_hip.so: multi-threaded, ~0.1 ms per
        1e6 float32 points on the GPU box's host; no optional Python dependency)."""
        if src.size == 0:
            return 0
        if src.flags.c_contiguous:
            return _capi.hash64(src)
        if src.flags.f_contiguous:                # (a transposed result, e.g. (R @ P.T).T: hash its buffer as it lies)
            return _capi.hash64(src.T) ^ 0x5bd1e995
        return _capi.hash64(np.ascontiguousarray(src))

    def _scan_for(self, source, fresh=False):
        """Upload (and Morton-sort) the scan; ``calc_H_g_e2`` called repeatedly with the same array
        (the Gauss-Newton pattern) reuses the device copy.  "Same" = same shape, dtype and content:
        the whole buffer is hashed on every call, so an in-place edit is always seen
        (``calc_H_g_e2`` stays pure in its inputs, as in the reference); ``align`` always uploads
        afresh."""
        if isinstance(source, UploadedScan):
            return source._scan
        src = np.asarray(source)
        if src.ndim != 2 or src.shape[1] != 3:
            raise ValueError("source must have shape (N, 3)")
        key = None if fresh else (src.shape, src.dtype.str, self._digest(src))
        if key is not None and self._scan is not None and self._scan_key == key:
            return self._scan
        if self._scan is not None:
            self._scan.close()
        self._scan = _capi.Scan(self._ctx(), src.astype(np.float32, copy=False))   # registration.py:83
        self._scan_key = key
        return self._scan

    def _linearize(self, cur_T, scan):
        out = _capi.linearize(self._target, scan, self.KIND, cur_T, self._max_dist(), self._call_flags())
        if self._needs_host_reduce():
            out = self._comm.allreduce(out)
        H, g, e2, cnt = _capi.unpack29(out)
        self.last_correspondences = cnt
        return H, g, e2

    def _set_target_handle(self, handle):
        if self._target is not None:
            self._target.close()
        self._target = handle
        self._is_target_set = True
