#!/usr/bin/env python3
"""Developer probe (round 5): what do the scan points with NO centroid inside the gate cost the voxel configs?  They are ~5 % of the
scan, interspersed (clutter above the ground), and each walks every ring that intersects the gate sphere -- with one such lane a
whole wave waits.  The same pass with those points removed / kept only / moved to the END of the scan (NO_SCAN_SORT keeps them together)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "vplane_10m"
kind_name, n_target, n_scan, vs, _ = bench.CONFIGS[cfg]
kind = {"vplane": _capi.VPLANE, "ndt": _capi.NDT}[kind_name]
ctx = _capi.get_context(0)
target = bench.make_cloud(n_target, seed=0)
scan, T_true = bench.make_scan(cfg, target, n_scan, seed=2)
tgt = _capi.Target.voxels(ctx, target, vs, 10)
st = (scan.astype(np.float64) @ T_true[:3, :3].T + T_true[:3, 3]).astype(np.float32)
d, i = tgt.nn_query(st, 2.0)
empty = i < 0
print(f"{cfg}: {int(empty.sum())} of {scan.shape[0]} scan points ({100 * empty.mean():.2f} %) have no centroid within the gate at T_true")
def timed(pts, tag, flags=0):
    sc = _capi.Scan(ctx, np.ascontiguousarray(pts), flags=flags)
    for P, name in ((np.eye(4), "identity"), (T_true, "T_true")):
        for _ in range(3):
            _capi.linearize(tgt, sc, kind, P, 2.0)
        ctx.profile_enable(True); ctx.profile_reset()
        for _ in range(8):
            out = _capi.linearize(tgt, sc, kind, P, 2.0)
        prof = ctx.profile_read(); ctx.profile_enable(False)
        print(f"{tag:<34} {pts.shape[0]:9d} points at {name:<9} " + ", ".join(f"{k} {v[1] / v[0] * 1e3:.1f} us" for k, v in prof.items() if v[0]) + f"; correspondences {int(out[28])}", flush=True)
    sc.close()
timed(scan, "whole scan")
timed(scan[~empty], "without the empty points")
timed(scan[empty], "the empty points alone")
