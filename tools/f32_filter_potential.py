#!/usr/bin/env python3
"""Developer probe: what a float32 filter in front of the float64 centroid search could save.  Runs the float32 POINT
search (plain, and tracking with a tiny margin -- what a filter-and-refine scheme would run) over the float32-rounded
centroids of a voxel config and prints its kernel time per pose next to the shipped float64 centroid search."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from point_cloud_registration_amd import _capi

cfg = sys.argv[1] if len(sys.argv) > 1 else "vplane_10m"
reps = 4
kind_name, n_target, n_scan, voxel_size, desc = B.CONFIGS[cfg]
kind = {"vplane": _capi.VPLANE, "ndt": _capi.NDT}[kind_name]
ctx = _capi.get_context(0)
target = B.make_cloud(n_target, seed=0)
scan, T_true = B.make_scan(cfg, target, n_scan, None, seed=2)
tv = _capi.Target.voxels(ctx, target, voxel_size, 10)
sc = _capi.Scan(ctx, scan)
T_fin, iters, trace = _capi.align(tv, sc, kind, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
traj = [trace[i, :16].reshape(4, 4).copy() for i in range(iters)]
means = tv.voxel_stats(("mean",))["mean"]
print(cfg, "centroids", means.shape, "index", tv.index_info())


def walk(tgt, k, label, mode, mu=None):
    ctx.set_reuse(mode, 0.0, mu if mu is not None else 0.0)
    t = np.zeros(len(traj))
    for r in range(reps + 1):
        for i, T in enumerate(traj):
            s2 = _capi.Scan(ctx, scan)          # fresh scan: a forced-reuse pass without history is a TRACK pass
            ctx.profile_enable(True); ctx.profile_reset()
            _capi.linearize(tgt, s2, k, T, 2.0)
            p = ctx.profile_read(); ctx.profile_enable(False)
            if r: t[i] += p["nn"][1] * 1e3 / reps
            st = s2.reuse_stats(); s2.close()
    print(f"{label:46s}", " ".join(f"{x:7.1f}" for x in t), f"| sum {t.sum():8.1f} us   last mode {'FTL'[st['last_mode']]}")


walk(tv, kind, "float64 centroid search (shipped)", 0)
m32 = np.ascontiguousarray(means.astype(np.float32))
for hint in (0.0, 2.0 * voxel_size, 1.5 * voxel_size, 3.0 * voxel_size):
    tp = _capi.Target.points(ctx, m32, None, cell_hint=hint)
    info = tp.index_info()
    walk(tp, _capi.ICP, f"float32 plain, cell {info['cell']:.2f} halo {info['halo']:.2f}", 0)
    walk(tp, _capi.ICP, f"float32 tracking mu=1e-3 cell, cell {info['cell']:.2f}", 2, 1e-3)
    tp.close()
ctx.set_reuse(1)
