#!/usr/bin/env python3
"""Developer probe: where the time of the class seam align(source-as-host-array) goes (upload, Morton sort, loop, frees)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import point_cloud_registration_amd as pcr
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, perturbed_scan
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_060_000
target = street(n, seed=0)
scan, _ = perturbed_scan(target, None, seed=2)
p = pcr.PlaneICP(max_dist=2.0, k=15); p.set_target(target)
ctx = _capi.get_context(0)


def med(fn, reps=15):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts[2:]) * 1e3


print(f"points {n}")
print(f"class align(array): {med(lambda: p.align(scan)):.3f} ms")
h = p.upload(scan)
print(f"class align(handle): {med(lambda: p.align(h)):.3f} ms")
print(f"upload() [create + drop]: {med(lambda: p.upload(scan)):.3f} ms")
keep = []
print(f"upload() [create, kept alive]: {med(lambda: keep.append(p.upload(scan))):.3f} ms")
del keep
sc = _capi.Scan(ctx, scan)
print(f"_capi.align(handle): {med(lambda: _capi.align(p._target, sc, _capi.PLANE, np.eye(4), 30, 1e-3, 2.0, want_trace=True)):.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    p.align(scan)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
os.environ["PCR_STALL_DEBUG"] = "0"
