#!/usr/bin/env python3
"""Per-pose NN search work counters along the bench trajectory (developer tool)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, perturbed_scan
target = street(1_060_000, seed=0); scan, _ = perturbed_scan(target, None)
ctx = _capi.get_context(0)
for cell in [float(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]:
    tgt = _capi.Target.points(ctx, target, cell_hint=cell); tgt.estimate_normals(15, want=False)
    sc = _capi.Scan(ctx, scan)
    T, it, tr = _capi.align(tgt, sc, 1, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
    print("cell", tgt.index_info()["cell"])
    for k in range(it):
        c = _capi.nn_counters(tgt, sc, tr[k, :16].reshape(4, 4), 2.0)
        print(f" pose {k}: " + " ".join(f"{a}={b:.2f}" for a, b in c.items()))
