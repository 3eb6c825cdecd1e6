#!/usr/bin/env python3
"""Developer probe: where the time of the class seam calc_H_g_e2(cur_T, source-as-array) goes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import point_cloud_registration_amd as pcr
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, perturbed_scan
target = street(1_060_000, seed=0)
scan, _ = perturbed_scan(target, None, seed=2)
p = pcr.PlaneICP(max_dist=2.0, k=15); p.set_target(target)
T = np.eye(4)
p.calc_H_g_e2(T, scan)
for name, fn in (("digest", lambda: p._digest(scan)), ("_scan_for", lambda: p._scan_for(scan)),
                 ("calc_H_g_e2(array)", lambda: p.calc_H_g_e2(T, scan))):
    s0 = p._scan
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms  (same device scan: {p._scan is s0})")
h = p.upload(scan)
t0 = time.perf_counter()
for _ in range(20):
    p.calc_H_g_e2(T, h)
print(f"calc_H_g_e2(handle): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    p.calc_H_g_e2(T, scan)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
