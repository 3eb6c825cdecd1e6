#!/usr/bin/env python3
"""Developer probe (run under rocprofv3 by tools/ta_probe.sh): 40 PlaneICP passes at ONE pose of the plane_b01
trajectory (argv[1] = pose index, default 0; argv[2] = "100m": the plane_100m workload instead) so that hardware
counters can be read per pose."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, perturbed_scan
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
big = len(sys.argv) > 2 and sys.argv[2] == "100m"        # the 1e8-point target with a 12.5 M-point scan
ctx = _capi.get_context(0)
ctx.set_reuse(0)                        # every pass a plain full search (40 passes at one pose would otherwise be certified)
if big:
    from point_cloud_registration_amd.synthetic import street_tiled
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
for _ in range(40):
    _capi.linearize(tgt, sc, _capi.PLANE, P, 2.0)
