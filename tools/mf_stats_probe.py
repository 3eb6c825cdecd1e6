#!/usr/bin/env python3
"""Developer probe (needs the PCR_MF_STATS build: tools/build_variant.sh mfstats "-DPCR_MF_STATS=1", PCR_LIB=build/exp/libpcr_mfstats.so):
work counters of the MFMA-filtered search per pose of a config's trajectory.   mf_stats_probe.py [config]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PCR_NN_MODE", "4")
from point_cloud_registration_amd import _capi
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "plane_b01"
kind_name, n_target, n_scan, voxel_size, desc = bench.CONFIGS[cfg]
ctx = _capi.get_context(0)
ctx.set_reuse(0)
print('blocks per CU (k_nn_scan<0>, <1>, coop, filter, mfma):', 'see nn_blocks_per_cu')
target = bench.make_cloud(n_target, seed=0)
scan, _ = bench.make_scan(cfg, target, n_scan, seed=2)
tgt = _capi.Target.points(ctx, target)
tgt.estimate_normals(15, compat=n_target <= 2_000_000, want=False)
sc = _capi.Scan(ctx, scan)
kind = {"icp": _capi.ICP, "plane": _capi.PLANE}[kind_name]
T, it, tr = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
L = _capi.lib()
L.pcr_mf_stats_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
names = ("query tiles", "boxed", "cand tiles", "rows", "rows swept", "lanes out", "lanes uncert", "list flushes")
for k in range(it):
    P = tr[k, :16].reshape(4, 4).copy()
    L.pcr_mf_stats_read(buf, 1)
    _capi.linearize(tgt, sc, kind, P, 2.0)
    ctx.synchronize()
    L.pcr_mf_stats_read(buf, 1)
    v = list(buf)
    q = max(v[0], 1)
    print(f"pose {k}: tiles {v[0]}, boxed {v[1] / q:.3f}, cand tiles/boxed {v[2] / max(v[1], 1):.1f}, rows/boxed {v[3] / max(v[1], 1):.1f} "
          f"(swept {v[4] / max(v[1], 1):.1f}), wave cycles per tile: seed {v[8] / q:.0f} box {v[9] / q:.0f} list {v[10] / q:.0f} sweep {v[11] / q:.0f} exact {v[12] / q:.0f} flush {v[13] / q:.0f} tile {v[14] / q:.0f}; lanes outside {v[5] / q:.2f}/tile, uncertified {v[6] / q:.3f}/tile, full-list flushes {v[7] / q:.4f}")
