#!/usr/bin/env python3
"""Developer probe: where the time of the class seam set_target(host array) goes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import point_cloud_registration_amd as pcr
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_060_000
target = street(n, seed=0)
ctx = _capi.get_context(0)


def med(fn, reps=12):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts[2:]) * 1e3


for name, mk in (("ICP", lambda: pcr.ICP()), ("PlaneICP", lambda: pcr.PlaneICP(k=15)), ("VPlaneICP", lambda: pcr.VPlaneICP(voxel_size=1.0)), ("NDT", lambda: pcr.NDT(voxel_size=1.0))):
    p = mk()
    print(f"{name}.set_target: {med(lambda: p.set_target(target)):.3f} ms")
print(f"Target.points (index only): {med(lambda: _capi.Target.points(ctx, target)):.3f} ms")
t = _capi.Target.points(ctx, target)
print(f"estimate_normals(15): {med(lambda: t.estimate_normals(15, compat=True, want=False)):.3f} ms")
import cProfile, pstats
p = pcr.PlaneICP(k=15)
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    p.set_target(target)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
