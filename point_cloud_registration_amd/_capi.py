"""ctypes binding of libpcr_hip.so (include/pcr.h) -- the only way the Python host reaches the GPU.

There is NO CPU fallback: if the HIP library is missing or no MI355X is visible, the calls
raise.  (The CPU restatement under oracle/ is test infrastructure and is never imported here.)
"""

import atexit
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCR_LIB: developer switch, another build of the same library: libpcr_hip_dev.so (`make dev`: + the developer / A-B
# kernels of csrc/kernels_dev.hip), or an experimental variant (tools/build_variant.sh)
LIB_PATH = os.environ.get("PCR_LIB") or os.path.join(_HERE, "libpcr_hip.so")
DEV_LIB_PATH = os.path.join(_HERE, "libpcr_hip_dev.so")

PCR_OK = 0
PCR_ERR_INVALID, PCR_ERR_HIP, PCR_ERR_NO_TARGET, PCR_ERR_COMM, PCR_ERR_SINGULAR, PCR_ERR_NOMEM = -1, -2, -3, -4, -5, -6
PCR_ERR_UNSUPPORTED = -7
ICP, PLANE, VPLANE, NDT = 0, 1, 2, 3
FLAG_ICP_RR_QUIRK = 1
FLAG_NO_SCAN_SORT = 2
FLAG_LOCAL_ONLY = 4          # no all-reduce even when the context has a communicator
FLAG_HOST_LOOP = 8           # pcr_align: host-driven loop instead of the device-resident one
FLAG_DEVICE_LOOP = 16        # pcr_align: device-resident loop even for small scans (where the host-driven one is picked)
K_LINEARIZE, K_FINALIZE, K_NN, K_REDUCE, K_ALLREDUCE, K_CERTIFY, K_COUNT = 0, 1, 2, 3, 4, 5, 6
KERNEL_NAMES = ("linearize", "finalize", "nn", "reduce", "allreduce", "certify")
NN_FULL, NN_TRACK, NN_LIST = 0, 1, 2      # what the search of a pass did (certified reuse, include/pcr.h)

_lib = None
_torch_lib_dir = None           # set when torch's bundled HIP runtime was pre-loaded (see below)
_live = weakref.WeakSet()      # targets / scans still holding device memory
_live_groups = weakref.WeakSet()   # pcr_groups (worker threads + member contexts)
_groups = {}
_shutdown = False

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_vp = C.c_void_p

# name -> (restype, argtypes); also the list of symbols the library must export
PROTOTYPES = {
    "pcr_last_error": (C.c_char_p, []),
    "pcr_version": (C.c_char_p, []),
    "pcr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pcr_context_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pcr_context_destroy": (C.c_int, [_vp]),
    "pcr_context_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "pcr_context_synchronize": (C.c_int, [_vp]),
    "pcr_comm_unique_id": (C.c_int, [C.c_char_p]),
    "pcr_comm_init": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    "pcr_comm_destroy": (C.c_int, [_vp]),
    "pcr_comm_p2p_export": (C.c_int, [_vp, C.c_char_p]),
    "pcr_comm_p2p_attach": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    "pcr_comm_p2p_failed": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "pcr_target_points_create": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_float, C.POINTER(_vp)]),
    "pcr_target_points_create_device": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_float, C.POINTER(_vp)]),
    "pcr_target_set_normals": (C.c_int, [_vp, _f32p]),
    "pcr_target_points_set_f64": (C.c_int, [_vp, _f64p]),
    "pcr_target_estimate_normals": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "pcr_target_get_normals": (C.c_int, [_vp, _f32p]),
    "pcr_target_voxels_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_double, C.c_int, C.POINTER(_vp)]),
    "pcr_target_voxels_create_from_stats": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_double, C.POINTER(_vp)]),
    "pcr_target_voxels_get": (C.c_int, [_vp, C.POINTER(C.c_int64), _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcr_target_size": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "pcr_target_destroy": (C.c_int, [_vp]),
    "pcr_scan_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_uint, C.POINTER(_vp)]),
    "pcr_scan_create_device": (C.c_int, [_vp, _vp, C.c_int64, C.c_uint, C.POINTER(_vp)]),
    "pcr_scan_size": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "pcr_scan_destroy": (C.c_int, [_vp]),
    "pcr_linearize": (C.c_int, [_vp, _vp, C.c_int, _f64p, C.c_double, C.c_uint, _f64p]),
    "pcr_align": (C.c_int, [_vp, _vp, C.c_int, _f64p, C.c_int, C.c_double, C.c_double, C.c_uint, _f64p,
                            C.POINTER(C.c_int), _vp]),
    "pcr_nn_query": (C.c_int, [_vp, _f32p, C.c_int64, C.c_float, _f32p, _i64p]),
    "pcr_nn_query_f64": (C.c_int, [_vp, _f32p, C.c_int64, C.c_double, _f64p, _i64p]),
    "pcr_knn_query": (C.c_int, [_vp, _f32p, C.c_int64, C.c_int, _f32p, _i64p]),
    "pcr_profile_enable": (C.c_int, [_vp, C.c_int]),
    "pcr_profile_reset": (C.c_int, [_vp]),
    "pcr_profile_read": (C.c_int, [_vp, _i64p, _f64p]),
    "pcr_target_index_info": (C.c_int, [_vp, C.POINTER(C.c_double), _i64p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pcr_target_index_halo": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pcr_target_index_population": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "pcr_set_variant": (C.c_int, [_vp, C.c_int]),
    "pcr_get_variant": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "pcr_set_nn_mode": (C.c_int, [_vp, C.c_int]),
    "pcr_set_fuse_finalize": (C.c_int, [_vp, C.c_int]),
    "pcr_get_pipeline": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pcr_nn_counters": (C.c_int, [_vp, _vp, _f64p, C.c_double, _f64p]),
    "pcr_set_reuse": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double]),
    "pcr_get_reuse": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pcr_scan_reuse_stats": (C.c_int, [_vp, _f64p]),
    "pcr_hash64": (C.c_int, [_vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pcr_target_filter_band": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "pcr_target_index_halo2": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pcr_usable_cpus": (C.c_int, []),
    "pcr_has_dev_kernels": (C.c_int, []),
    "pcr_abi_version": (C.c_int, []),
    "pcr_context_trim": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "pcr_profile_read_n": (C.c_int, [_vp, C.c_int, _i64p, _f64p, C.POINTER(C.c_int)]),
    "pcr_scan_read_matches": (C.c_int, [_vp, _vp]),
    "pcr_comm_p2p_finegrained": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "pcr_lzf_decompress": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pcr_lzf_compress": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    # single-process multi-device groups (include/pcr.h)
    "pcr_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]),
    "pcr_group_destroy": (C.c_int, [_vp]),
    "pcr_group_size": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "pcr_group_context": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "pcr_group_target_points_create": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_float, C.POINTER(_vp)]),
    "pcr_group_target_voxels_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_double, C.c_int, C.POINTER(_vp)]),
    "pcr_group_target_estimate_normals": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "pcr_group_target_set_normals": (C.c_int, [_vp, _f32p]),
    "pcr_group_target_points_set_f64": (C.c_int, [_vp, _f64p]),
    "pcr_group_target_member": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "pcr_group_target_destroy": (C.c_int, [_vp]),
    "pcr_group_scan_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_uint, C.POINTER(_vp)]),
    "pcr_group_scan_size": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "pcr_group_scan_destroy": (C.c_int, [_vp]),
    "pcr_group_linearize": (C.c_int, [_vp, _vp, C.c_int, _f64p, C.c_double, C.c_uint, _f64p]),
    "pcr_group_align": (C.c_int, [_vp, _vp, C.c_int, _f64p, C.c_int, C.c_double, C.c_double, C.c_uint, _f64p,
                                  C.POINTER(C.c_int), _vp]),
}
ABI_VERSION = 5                 # PCR_ABI_VERSION of the include/pcr.h this binding was written against


class PcrError(RuntimeError):
    """HIP / RCCL / argument failure reported by libpcr_hip.so."""


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so and loads it
    by path; if libpcr_hip.so has already pulled in /opt/rocm's copy, a later ``import torch`` leaves
    TWO runtimes in the process (observed: "double free or corruption" at exit).  When torch is
    installed but not imported yet, load its runtime first so both sides resolve to the same one
    (same SONAME, libamdhip64.so.7).  PCR_KEEP_SYSTEM_HIP=1 skips this.

    Everything is loaded with RTLD_LOCAL: the loader reuses an already-loaded object by SONAME / inode
    whatever its scope, and putting RCCL's dependency librocm_smi64.so into the GLOBAL scope makes a
    later libamd_smi.so (loaded by ``import torch`` through the amdsmi module) bind its static
    ``amd::smi`` maps to librocm_smi64's copies -- constructed twice, destroyed twice, glibc aborts
    at exit (reproducible with six lines of ctypes, no libpcr involved)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("PCR_KEEP_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    global _torch_lib_dir
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    path = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_LOCAL)
            _torch_lib_dir = libdir
        except OSError:
            pass


def _share_rccl_with_torch():
    """Same idea for RCCL (bound lazily with dlopen("librccl.so.1") inside libpcr_hip.so): when the
    process runs on torch's HIP runtime, use the RCCL built against it."""
    if _torch_lib_dir is None:
        return
    path = os.path.join(_torch_lib_dir, "librccl.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_LOCAL)
        except OSError:
            pass


def lib():
    """Load libpcr_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PcrError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the registration hot path.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(L, name)      # AttributeError if the symbol is not exported
        except AttributeError:
            if not os.environ.get("PCR_LIB"):
                raise
            if name == "pcr_abi_version":
                raise PcrError(f"{LIB_PATH} predates pcr_abi_version (ABI < 4): too old for this binding, rebuild it")
            # an A/B library built from an older revision (tools/build_rev_lib.sh): instrumentation entry points it
            # does not have yet read as "nothing" (status 0, outputs untouched)
            setattr(L, name, lambda *a, **k: 0)
            continue
        fn.restype = res
        fn.argtypes = args
    if L.pcr_abi_version() != ABI_VERSION:
        raise PcrError(f"{LIB_PATH} has ABI version {L.pcr_abi_version()}, this binding expects {ABI_VERSION}: rebuild it")
    _lib = L
    return L


def check(status):
    if status == PCR_OK:
        return
    msg = lib().pcr_last_error().decode("utf-8", "replace")
    if status == PCR_ERR_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")          # what numpy.linalg.solve raises (quirk Q7)
    if status == PCR_ERR_INVALID:
        raise ValueError(msg)
    if status == PCR_ERR_NO_TARGET:
        raise ValueError(msg)
    raise PcrError(f"libpcr_hip status {status}: {msg}")


def has_dev_kernels():
    """True when the loaded library is a developer build (unfused folds, wave-cooperative search, work counters)."""
    return bool(lib().pcr_has_dev_kernels())


def device_count():
    n = C.c_int(0)
    check(lib().pcr_device_count(C.byref(n)))
    return n.value


class Context:
    """One GPU + one HIP stream (pcr_context)."""

    def __init__(self, device=0):
        self.device = int(device)
        h = _vp()
        check(lib().pcr_context_create(self.device, C.byref(h)))
        self.handle = h
        self.nranks, self.rank = 1, 0

    def synchronize(self):
        check(lib().pcr_context_synchronize(self.handle))

    def stream(self):
        s = _vp()
        check(lib().pcr_context_stream(self.handle, C.byref(s)))
        return s.value

    def set_variant(self, v):
        check(lib().pcr_set_variant(self.handle, int(v)))

    def get_variant(self):
        v = C.c_int(0)
        check(lib().pcr_get_variant(self.handle, C.byref(v)))
        return v.value

    def set_nn_mode(self, m):
        check(lib().pcr_set_nn_mode(self.handle, int(m)))

    def set_fuse_finalize(self, on):
        check(lib().pcr_set_fuse_finalize(self.handle, int(bool(on))))

    def set_reuse(self, mode=None, tau=0.0, mu=0.0):
        """Certified reuse of the previous pass' matches: 0 off (default), 1 automatic, 2 always; ``tau`` / ``mu``
        in units of the target index' cell size (<= 0 keeps the current value)."""
        if mode is None:
            mode = self.get_reuse()["mode"]
        check(lib().pcr_set_reuse(self.handle, int(mode), float(tau), float(mu)))

    def get_reuse(self):
        m, t, u = C.c_int(0), C.c_double(0), C.c_double(0)
        check(lib().pcr_get_reuse(self.handle, C.byref(m), C.byref(t), C.byref(u)))
        return {"mode": m.value, "tau": t.value, "mu": u.value}

    def get_pipeline(self):
        v, f, m = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().pcr_get_pipeline(self.handle, C.byref(v), C.byref(f), C.byref(m)))
        return {"variant": v.value, "fuse_finalize": f.value, "nn_mode": m.value, "reuse": self.get_reuse()["mode"]}

    def pipeline(self, variant=None, fuse_finalize=None, nn_mode=None, reuse=None):
        """Context manager: select a kernel pipeline for the enclosed calls and RESTORE the previous
        selection afterwards (the context is process-wide)."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            prev = self.get_pipeline()
            try:
                if variant is not None:
                    self.set_variant(variant)
                if fuse_finalize is not None:
                    self.set_fuse_finalize(fuse_finalize)
                if nn_mode is not None:
                    self.set_nn_mode(nn_mode)
                if reuse is not None:
                    self.set_reuse(reuse)
                yield self
            finally:
                # restore everything that can be restored; a failing setter must not mask the body's own exception
                # or leave the rest of the selection behind
                errs = []
                for fn, val in ((self.set_variant, prev["variant"]), (self.set_fuse_finalize, prev["fuse_finalize"]),
                                (self.set_nn_mode, prev["nn_mode"]), (self.set_reuse, prev["reuse"])):
                    try:
                        fn(val)
                    except Exception as exc:          # noqa: BLE001
                        errs.append(exc)
                import sys as _sys
                if errs and _sys.exc_info()[0] is None:
                    raise errs[0]
        return _cm()

    # -- RCCL
    def comm_init(self, uid, nranks, rank):
        _share_rccl_with_torch()
        check(lib().pcr_comm_init(self.handle, uid, int(nranks), int(rank)))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_p2p_export(self):
        """Peer-to-peer transport (include/pcr.h): this rank's 64-byte IPC handle, to be all-gathered out of band."""
        buf = C.create_string_buffer(64)
        check(lib().pcr_comm_p2p_export(self.handle, buf))
        return buf.raw

    def comm_p2p_attach(self, handles, rank):
        blob = b"".join(handles)
        assert len(blob) == 64 * len(handles)
        check(lib().pcr_comm_p2p_attach(self.handle, blob, len(handles), int(rank)))
        self.nranks, self.rank = len(handles), int(rank)

    def comm_p2p_finegrained(self):
        """True: this rank's slots are fine-grained device memory (coherent across devices while a kernel runs)."""
        f = C.c_int(0)
        check(lib().pcr_comm_p2p_finegrained(self.handle, C.byref(f)))
        return bool(f.value)

    def comm_p2p_failed(self):
        f = C.c_int(0)
        check(lib().pcr_comm_p2p_failed(self.handle, C.byref(f)))
        return bool(f.value)

    def comm_destroy(self):
        check(lib().pcr_comm_destroy(self.handle))
        self.nranks, self.rank = 1, 0

    # -- profiling
    def profile_enable(self, on=True, period=1):
        """HIP events around the hot-path launches; ``period`` = n brackets every n-th pass only."""
        check(lib().pcr_profile_enable(self.handle, (max(int(period), 1) if on else 0)))

    def profile_reset(self):
        check(lib().pcr_profile_reset(self.handle))

    def profile_read(self):
        n = np.zeros(K_COUNT, np.int64)
        ms = np.zeros(K_COUNT, np.float64)
        cnt = C.c_int(0)
        check(lib().pcr_profile_read_n(self.handle, K_COUNT, n, ms, C.byref(cnt)))
        return {KERNEL_NAMES[i]: (int(n[i]), float(ms[i])) for i in range(min(K_COUNT, cnt.value))}

    def trim(self):
        """hipFree every idle block of the context's block cache (destroyed targets / scans leave up to 1 GiB there,
        invisible to torch's allocator); returns the bytes released."""
        b = C.c_uint64(0)
        check(lib().pcr_context_trim(self.handle, C.byref(b)))
        return int(b.value)

    def close(self):
        if getattr(self, "handle", None) and not _shutdown and not getattr(self, "_borrowed", False):
            lib().pcr_context_destroy(self.handle)
        self.handle = None


@atexit.register
def _release_all():
    """Free every device object and context while the HIP runtime is still alive.  Leaving this to
    ``__del__`` during interpreter teardown runs hipFree after the runtime's own static destructors
    (observed: "double free or corruption" at exit of a long pytest run)."""
    global _shutdown
    for obj in list(_live):
        try:
            obj.close()
        except Exception:
            pass
    for ctx in list(_contexts.values()):
        try:
            ctx.close()
        except Exception:
            pass
    _contexts.clear()
    for grp in list(_groups.values()) + list(_live_groups):
        try:
            grp.close()
        except Exception:
            pass
    _groups.clear()
    _shutdown = True


def comm_unique_id():
    buf = C.create_string_buffer(128)
    _share_rccl_with_torch()
    check(lib().pcr_comm_unique_id(buf))
    return buf.raw


_contexts = {}


def get_context(device=None):
    """Process-wide context per device; default device = LOCAL_RANK (one process per GPU)."""
    if device is None:
        device = int(os.environ.get("PCR_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = device_count()
        if n > 0:
            device %= n
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


class Target:
    """pcr_target handle (point target or voxel target)."""

    def __init__(self, ctx, handle, is_voxel):
        self.ctx, self.handle, self.is_voxel = ctx, handle, is_voxel
        _live.add(self)

    @classmethod
    def points(cls, ctx, xyz, normals=None, cell_hint=0.0):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if normals is not None:
            normals = np.ascontiguousarray(normals, dtype=np.float32)
            if normals.shape != xyz.shape:
                raise ValueError("normals must have the shape of the target")
        h = _vp()
        if isinstance(ctx, Group):
            check(lib().pcr_group_target_points_create(ctx.handle, _ptr(xyz), xyz.shape[0], _ptr(normals),
                                                       float(cell_hint), C.byref(h)))
            return GroupTarget(ctx, h, False)
        check(lib().pcr_target_points_create(ctx.handle, _ptr(xyz), xyz.shape[0], _ptr(normals),
                                             float(cell_hint), C.byref(h)))
        return cls(ctx, h, False)

    @classmethod
    def points_device(cls, ctx, d_xyz, n, d_normals=None, cell_hint=0.0):
        h = _vp()
        check(lib().pcr_target_points_create_device(ctx.handle, _vp(d_xyz), int(n),
                                                    _vp(d_normals) if d_normals else None,
                                                    float(cell_hint), C.byref(h)))
        return cls(ctx, h, False)

    @classmethod
    def voxels(cls, ctx, xyz, voxel_size, min_points=10):
        a = np.asarray(xyz)
        is64 = a.dtype == np.float64
        a = np.ascontiguousarray(a, dtype=np.float64 if is64 else np.float32)
        h = _vp()
        if isinstance(ctx, Group):
            check(lib().pcr_group_target_voxels_create(ctx.handle, _ptr(a), int(is64), a.shape[0], float(voxel_size),
                                                       int(min_points), C.byref(h)))
            return GroupTarget(ctx, h, True)
        check(lib().pcr_target_voxels_create(ctx.handle, _ptr(a), int(is64), a.shape[0], float(voxel_size),
                                             int(min_points), C.byref(h)))
        return cls(ctx, h, True)

    @classmethod
    def voxels_from_stats(cls, ctx, mean, norm, icov, voxel_size):
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        norm = None if norm is None else np.ascontiguousarray(norm, dtype=np.float64)
        icov = None if icov is None else np.ascontiguousarray(icov, dtype=np.float64)
        h = _vp()
        check(lib().pcr_target_voxels_create_from_stats(ctx.handle, _ptr(mean), _ptr(norm), _ptr(icov),
                                                        mean.shape[0], float(voxel_size), C.byref(h)))
        return cls(ctx, h, True)

    def size(self):
        n = C.c_int64(0)
        check(lib().pcr_target_size(self.handle, C.byref(n)))
        return n.value

    def set_points_f64(self, xyz64):
        """Quirk Q6 (plane_icp.py:20-22, kdtree.py:18-21): the float64 coordinates of the points this target was created
        from -- PlaneICP passes and ``nn_query`` then search in float64, as the reference's tree over a float64 array does."""
        xyz64 = np.ascontiguousarray(xyz64, dtype=np.float64)
        if xyz64.shape != (self.size(), 3):
            raise ValueError("xyz64 must have the shape of the target")
        fn = lib().pcr_group_target_points_set_f64 if getattr(self, "ghandle", None) else lib().pcr_target_points_set_f64
        if not hasattr(fn, "argtypes"):          # (an older PCR_LIB without the entry point: its stub would "succeed")
            return False
        st = fn(getattr(self, "ghandle", None) or self.handle, xyz64)
        if st == PCR_ERR_UNSUPPORTED:
            # coordinates float32 cannot resolve (UTM-scale clouds): the float32 index stays, as before Q6 was reproduced
            import warnings
            warnings.warn("float64 target searched through its float32 copy: "
                          + lib().pcr_last_error().decode("utf-8", "replace"), RuntimeWarning, stacklevel=3)
            return False
        check(st)
        self.has_f64 = True
        return True

    def set_normals(self, normals):
        check(lib().pcr_target_set_normals(self.handle, np.ascontiguousarray(normals, dtype=np.float32)))

    def estimate_normals(self, k=15, compat=True, want=True):
        out = np.empty((self.size(), 3), np.float32) if want else None
        check(lib().pcr_target_estimate_normals(self.handle, int(k), int(bool(compat)), _ptr(out)))
        return out

    def get_normals(self):
        out = np.empty((self.size(), 3), np.float32)
        check(lib().pcr_target_get_normals(self.handle, out))
        return out

    def voxel_stats(self, names=("mean", "cov", "norm", "icov", "counts", "keys")):
        n = self.size()
        bufs = {"mean": np.empty((n, 3)), "cov": np.empty((n, 3, 3)), "norm": np.empty((n, 3)),
                "icov": np.empty((n, 3, 3)), "counts": np.empty(n, np.int64), "keys": np.empty(n, np.int64)}
        nv = C.c_int64(0)
        args = [_ptr(bufs[k]) if k in names else None for k in ("mean", "cov", "norm", "icov", "counts", "keys")]
        check(lib().pcr_target_voxels_get(self.handle, C.byref(nv), *args))
        return {k: bufs[k] for k in names}

    def index_info(self):
        cell, occ, n = C.c_double(0), C.c_int64(0), C.c_int64(0)
        dims = np.zeros(3, np.int64)
        check(lib().pcr_target_index_info(self.handle, C.byref(cell), dims, C.byref(occ), C.byref(n)))
        halo, nh = C.c_double(0), C.c_int64(0)
        check(lib().pcr_target_index_halo(self.handle, C.byref(halo), C.byref(nh)))
        band = C.c_double(0)
        check(lib().pcr_target_filter_band(self.handle, C.byref(band)))
        halo2, nh2 = C.c_double(0), C.c_int64(0)
        check(lib().pcr_target_index_halo2(self.handle, C.byref(halo2), C.byref(nh2)))
        pmax, p99, heavy = C.c_int64(0), C.c_int64(0), C.c_int(0)
        check(lib().pcr_target_index_population(self.handle, C.byref(pmax), C.byref(p99), C.byref(heavy)))
        return {"cell": cell.value, "dims": tuple(int(d) for d in dims), "occupied": occ.value, "n": n.value,
                "halo": halo.value, "halo_records": nh.value, "filter_band": band.value,
                "halo2": halo2.value, "halo2_records": nh2.value,
                "pop_max": pmax.value, "pop_p99": p99.value, "heavy": bool(heavy.value)}

    def nn_query(self, q, r_max=np.inf):
        q = np.ascontiguousarray(q, dtype=np.float32)
        m = q.shape[0]
        idx = np.empty(m, np.int64)
        if self.is_voxel or getattr(self, "has_f64", False):
            dist = np.empty(m, np.float64)
            check(lib().pcr_nn_query_f64(self.handle, q, m, float(r_max), dist, idx))
        else:
            dist = np.empty(m, np.float32)
            check(lib().pcr_nn_query(self.handle, q, m, float(r_max), dist, idx))
        return dist, idx

    def knn_query(self, q, k):
        q = np.ascontiguousarray(q, dtype=np.float32)
        m = q.shape[0]
        dist = np.empty((m, k), np.float32)
        idx = np.empty((m, k), np.int64)
        check(lib().pcr_knn_query(self.handle, q, m, int(k), dist, idx))
        return dist, idx

    def close(self):
        if getattr(self, "handle", None) and not _shutdown:
            lib().pcr_target_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scan:
    """pcr_scan handle: the float32 scan, uploaded and Morton-sorted once per align()."""

    def __init__(self, ctx, xyz=None, flags=0, device_ptr=None, n=None):
        self.ctx = ctx
        self.ghandle = None
        h = _vp()
        if isinstance(ctx, Group):
            # a group scan: cut into one contiguous shard per member inside pcr_group_scan_create
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            if xyz.ndim != 2 or xyz.shape[1] != 3:
                raise ValueError("scan must have shape (N, 3)")
            check(lib().pcr_group_scan_create(ctx.handle, _ptr(xyz), xyz.shape[0], int(flags), C.byref(h)))
            self.n = xyz.shape[0]
            self.ghandle, self.handle = h, None
            _live.add(self)
            return
        if device_ptr is not None:
            check(lib().pcr_scan_create_device(ctx.handle, _vp(device_ptr), int(n), int(flags), C.byref(h)))
            self.n = int(n)
        else:
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            if xyz.ndim != 2 or xyz.shape[1] != 3:
                raise ValueError("scan must have shape (N, 3)")
            check(lib().pcr_scan_create(ctx.handle, _ptr(xyz), xyz.shape[0], int(flags), C.byref(h)))
            self.n = xyz.shape[0]
        self.handle = h
        _live.add(self)

    def reuse_stats(self):
        """Certified reuse on this scan: passes by search mode, points the list passes searched, last pass."""
        o = np.zeros(8)
        check(lib().pcr_scan_reuse_stats(self.handle, o))
        return {"passes_full": int(o[0]), "passes_track": int(o[1]), "passes_list": int(o[2]),
                "list_searched": int(o[3]), "list_points": int(o[4]),
                "last_mode": int(o[5]), "last_searched": int(o[6]), "last_motion": float(o[7])}

    def matches(self):
        """The correspondences of the last search + reduce pass over this scan (test / diagnostic seam): per scan
        point in the scan's DEVICE order, the cell-sorted index of the matched target record, -1 = none."""
        out = np.empty(self.n, np.uint32)
        check(lib().pcr_scan_read_matches(self.handle, _ptr(out)))
        m = out.astype(np.int64)
        m[out == 0xFFFFFFFF] = -1
        return m

    def close(self):
        if getattr(self, "ghandle", None) and not _shutdown:
            lib().pcr_group_scan_destroy(self.ghandle)
        elif getattr(self, "handle", None) and not _shutdown:
            lib().pcr_scan_destroy(self.handle)
        self.handle = None
        self.ghandle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    """pcr_group: single-process multi-device (include/pcr.h).  One context + host thread per entry of ``devices`` (an id
    may repeat: ``[0, 0]`` = two contexts on one GPU).  Pass it wherever a Context is expected by ``Target.points`` /
    ``Target.voxels`` / ``Scan``: targets are then built on every member, scans sharded contiguously, and ``linearize`` /
    ``align`` return the sums / the pose over the WHOLE scan."""

    def __init__(self, devices):
        self.devices = tuple(int(d) for d in devices)
        if not 1 <= len(self.devices) <= 8:
            raise ValueError("a group has 1 to 8 member devices")
        ids = (C.c_int * len(self.devices))(*self.devices)
        h = _vp()
        check(lib().pcr_group_create(ids, len(self.devices), C.byref(h)))
        self.handle = h
        self.nranks, self.rank = len(self.devices), 0
        _live_groups.add(self)

    def member(self, i):
        """Member i's context (borrowed, not owned): profiling, pipeline switches, synchronize."""
        h = _vp()
        check(lib().pcr_group_context(self.handle, int(i), C.byref(h)))
        c = Context.__new__(Context)
        c.device, c.handle, c.nranks, c.rank, c._borrowed = self.devices[i], h, len(self.devices), int(i), True
        return c

    def synchronize(self):
        for i in range(len(self.devices)):
            self.member(i).synchronize()

    def close(self):
        if getattr(self, "handle", None) and not _shutdown:
            lib().pcr_group_destroy(self.handle)
        self.handle = None


def get_group(devices):
    """Process-wide group per device tuple (like get_context per device)."""
    key = tuple(int(d) for d in devices)
    if key not in _groups:
        _groups[key] = Group(key)
    return _groups[key]


class GroupTarget(Target):
    """pcr_group_target: the same target on every member.  ``handle`` is member 0's pcr_target (borrowed), so every
    read-only method of Target (statistics, queries, index_info) answers from member 0; what changes the target runs on all."""

    def __init__(self, group, ghandle, is_voxel):
        self.ghandle = ghandle
        h = _vp()
        check(lib().pcr_group_target_member(ghandle, 0, C.byref(h)))
        super().__init__(group, h, is_voxel)

    def set_normals(self, normals):
        check(lib().pcr_group_target_set_normals(self.ghandle, np.ascontiguousarray(normals, dtype=np.float32)))

    def estimate_normals(self, k=15, compat=True, want=True):
        out = np.empty((self.size(), 3), np.float32) if want else None
        check(lib().pcr_group_target_estimate_normals(self.ghandle, int(k), int(bool(compat)), _ptr(out)))
        return out

    def close(self):
        if getattr(self, "ghandle", None) and not _shutdown:
            lib().pcr_group_target_destroy(self.ghandle)
        self.ghandle = None
        self.handle = None


def lzf_decompress(data, out_len):
    """LZF stream -> bytes of length out_len (pcr_lzf_decompress: host C, no GPU needed)."""
    src = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(int(out_len), np.uint8)
    w = C.c_uint64(0)
    st = lib().pcr_lzf_decompress(src.ctypes.data_as(_vp), src.size, out.ctypes.data_as(_vp), out.size, C.byref(w))
    if st != PCR_OK or w.value != out_len:
        raise ValueError("corrupt LZF stream in PCD file")
    return out.tobytes()


def lzf_compress(data):
    src = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(src.size + src.size // 32 + 16, np.uint8)
    w = C.c_uint64(0)
    check(lib().pcr_lzf_compress(src.ctypes.data_as(_vp), src.size, out.ctypes.data_as(_vp), out.size, C.byref(w)))
    return out[:w.value].tobytes()


def hash64(arr):
    """Content hash of a C-contiguous NumPy array's buffer (pcr_hash64: multi-threaded, 64 bits)."""
    h = C.c_uint64(0)
    check(lib().pcr_hash64(arr.ctypes.data_as(_vp), arr.nbytes, C.byref(h)))
    return h.value


_linearize_fast = None


def linearize(target, scan, kind, T, max_dist, flags=FLAG_ICP_RR_QUIRK):
    """One pass of the hot path -> the 29 sums (see include/pcr.h).

    Called once per Gauss-Newton iteration: the binding goes through a second ctypes prototype with plain pointer
    arguments (``numpy.ctypeslib.ndpointer`` re-validates dtype / flags / shape of every array on every call, a few
    microseconds against a 50-150 us pass); the arrays are made float64 C-contiguous right here."""
    global _linearize_fast
    if _linearize_fast is None:
        proto = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_int, _vp, C.c_double, C.c_uint, _vp)
        _linearize_fast = C.cast(lib().pcr_linearize, proto)
    out = np.empty(29)
    if not (isinstance(T, np.ndarray) and T.dtype == np.float64 and T.flags.c_contiguous and T.size == 16):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    if getattr(target, "ghandle", None) is not None:              # a group: the sums over every member's shard
        if getattr(scan, "ghandle", None) is None:
            raise ValueError("a group target needs a scan created on the same group")
        check(lib().pcr_group_linearize(target.ghandle, scan.ghandle, int(kind), T.reshape(16), float(max_dist), int(flags), out))
        return out
    st = _linearize_fast(target.handle, scan.handle, kind, T.ctypes.data, max_dist, flags, out.ctypes.data)
    if st != PCR_OK:
        check(st)
    return out


def align(target, scan, kind, T_init, max_iter, tol, max_dist, flags=FLAG_ICP_RR_QUIRK, want_trace=False):
    """pcr_align: the whole Gauss-Newton loop behind the boundary."""
    T0 = np.ascontiguousarray(T_init, dtype=np.float64).reshape(16)
    T = np.zeros(16)
    iters = C.c_int(0)
    trace = np.zeros((max(int(max_iter), 1), 45)) if want_trace else None
    if getattr(target, "ghandle", None) is not None:
        if getattr(scan, "ghandle", None) is None:
            raise ValueError("a group target needs a scan created on the same group")
        check(lib().pcr_group_align(target.ghandle, scan.ghandle, int(kind), T0, int(max_iter), float(tol), float(max_dist),
                                    int(flags), T, C.byref(iters), _ptr(trace)))
    else:
        check(lib().pcr_align(target.handle, scan.handle, int(kind), T0, int(max_iter), float(tol), float(max_dist),
                              int(flags), T, C.byref(iters), _ptr(trace)))
    T = T.reshape(4, 4)
    if want_trace:
        return T, iters.value, trace[:iters.value]
    return T, iters.value


def nn_counters(target, scan, T, max_dist):
    """Search work counters for one pose (see include/pcr.h: pcr_nn_counters)."""
    out = np.zeros(11)
    check(lib().pcr_nn_counters(target.handle, scan.handle, np.ascontiguousarray(T, np.float64).reshape(16),
                                float(max_dist), out))
    names = ("rings", "rows_loaded", "rows_pruned", "candidates")
    n = max(scan.n, 1)
    waves = max((scan.n + 63) // 64, 1)
    return {**{k: out[i] / n for i, k in enumerate(names)},
            **{"wave_" + k: out[4 + i] / n for i, k in enumerate(names)},
            "cyc_prologue": out[8] / waves, "cyc_ring0": out[9] / waves, "cyc_rings": out[10] / waves}


def unpack29(out):
    """29 sums -> (H 6x6 symmetric, g 6, e2, count)."""
    H = np.zeros((6, 6))
    H[np.triu_indices(6)] = out[:21]
    H = H + np.triu(H, 1).T
    return H, np.array(out[21:27]), float(out[27]), int(round(out[28]))
