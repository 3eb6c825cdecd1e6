#!/usr/bin/env python3
"""set_target-side timings (index build, normals, voxel build) at several sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, street_tiled
ctx = _capi.get_context(0)
w = street(20000)                      # warm-up: first use of every build kernel (code load) stays out of the timings
_capi.Target.points(ctx, w).estimate_normals(15, want=False); _capi.Target.voxels(ctx, w, 1.0, 10).close(); _capi.Scan(ctx, w).close()
for n in [int(float(a)) for a in (sys.argv[1:] or ["1.06e6", "1e7"])]:
    pts = street(n) if n <= 2_000_000 else street_tiled(n)
    t0 = time.perf_counter(); t = _capi.Target.points(ctx, pts); ctx.synchronize(); t1 = time.perf_counter()
    t.estimate_normals(15, want=False); ctx.synchronize(); t2 = time.perf_counter()
    info = t.index_info(); t.close()
    t3 = time.perf_counter(); v = _capi.Target.voxels(ctx, pts, 1.0, 10); ctx.synchronize(); t4 = time.perf_counter()
    nv = v.size(); v.close()
    sc0 = time.perf_counter(); s = _capi.Scan(ctx, pts[: min(n, 12_500_000)]); ctx.synchronize(); sc1 = time.perf_counter(); s.close()
    print(f"n={n}: point index {1e3 * (t1 - t0):.2f} ms (cell {info['cell']:.3f}, dims {info['dims']}, halo records {info['halo_records']}), "
          f"normals k=15 {1e3 * (t2 - t1):.2f} ms, voxel build {1e3 * (t4 - t3):.2f} ms ({nv} voxels), "
          f"scan upload+sort {1e3 * (sc1 - sc0):.2f} ms", flush=True)
