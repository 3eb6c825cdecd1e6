#!/usr/bin/env python3
"""Developer probe: N PlaneICP passes at ONE pose of the plane_b01 (or, argv[2] = "100m", plane_100m) trajectory, kernel times
from the library's own HIP events (profile_read).   pose_passes_timed.py <pose> [100m] [passes]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, street_tiled, perturbed_scan
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
big = len(sys.argv) > 2 and sys.argv[2] == "100m"
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ctx = _capi.get_context(0)
ctx.set_reuse(0)
if big:
    target = street_tiled(100_000_000, seed=0)
    tgt = _capi.Target.points(ctx, target); tgt.estimate_normals(15, compat=False, want=False)
    scan, _ = perturbed_scan(target, 12_500_000, seed=2)
else:
    target = street(1_060_000, seed=0)
    tgt = _capi.Target.points(ctx, target); tgt.estimate_normals(15, want=False)
    scan, _ = perturbed_scan(target, None, seed=2)
sc = _capi.Scan(ctx, scan)
T, it, tr = _capi.align(tgt, sc, _capi.PLANE, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
P = tr[min(pose, it - 1), :16].reshape(4, 4).copy()
for _ in range(16):
    ref = _capi.linearize(tgt, sc, _capi.PLANE, P, 2.0)
ctx.profile_enable(True)
ctx.profile_reset()
for _ in range(passes):
    out = _capi.linearize(tgt, sc, _capi.PLANE, P, 2.0)
prof = ctx.profile_read()
ctx.profile_enable(False)
print(f"pose {pose}: sums identical to the warm-up pass: {np.array_equal(out, ref)};", {k: f"{ms / n * 1000:.1f} us x {n}" for k, (n, ms) in prof.items() if n})
