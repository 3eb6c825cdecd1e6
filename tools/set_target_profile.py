#!/usr/bin/env python3
"""Developer probe for the set_target side (rows N1 / N2 / T4 of SURVEY.md section 8): repeated builds of the point index,
the k = 15 PCA normals, the voxel target and a scan at one size, so that `rocprofv3 --kernel-trace --stats` (and the
--pmc FETCH_SIZE / WRITE_SIZE passes) of this command give per-kernel averages.  Prints host wall-clock medians too.
    python tools/set_target_profile.py <n points> [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, street_tiled
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_060_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = _capi.get_context(0)
pts = street(n) if n <= 2_000_000 else street_tiled(n)
w = street(20000)
_capi.Target.points(ctx, w).estimate_normals(15, want=False); _capi.Target.voxels(ctx, w, 1.0, 10).close(); _capi.Scan(ctx, w).close()
T = {"index": [], "normals": [], "voxels1.0": [], "voxels0.5": [], "scan": []}
for r in range(reps):
    t0 = time.perf_counter(); t = _capi.Target.points(ctx, pts); ctx.synchronize(); T["index"].append(time.perf_counter() - t0)
    t0 = time.perf_counter(); t.estimate_normals(15, compat=n <= 2_000_000, want=False); ctx.synchronize(); T["normals"].append(time.perf_counter() - t0)
    t.close()
    for vs in (1.0, 0.5):
        t0 = time.perf_counter(); v = _capi.Target.voxels(ctx, pts, vs, 10); ctx.synchronize(); T[f"voxels{vs}"].append(time.perf_counter() - t0)
        v.close()
    t0 = time.perf_counter(); s = _capi.Scan(ctx, pts[: min(n, 12_500_000)]); ctx.synchronize(); T["scan"].append(time.perf_counter() - t0)
    s.close()
print(f"n={n} reps={reps} host wall ms (median of the last {max(reps - 2, 1)}): " +
      ", ".join(f"{k} {1e3 * float(np.median(v[2:] or v)):.3f}" for k, v in T.items()), flush=True)
