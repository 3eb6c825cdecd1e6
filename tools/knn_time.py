#!/usr/bin/env python3
"""Developer probe: k-NN normals wall time (median of reps) for the library PCR_LIB points at.   knn_time.py <n> [k ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, street_tiled
n = int(float(sys.argv[1]))
ks = [int(a) for a in sys.argv[2:]] or [15]
ctx = _capi.get_context(0)
pts = street(n) if n <= 2_000_000 else street_tiled(n)
t = _capi.Target.points(ctx, pts)
out = []
for k in ks:
    ts = []
    for r in range(8 if n <= 2_000_000 else 4):
        t0 = time.perf_counter(); t.estimate_normals(k, compat=n <= 2_000_000, want=False); ctx.synchronize(); ts.append(time.perf_counter() - t0)
    out.append(f"k={k} {1e3 * float(np.median(ts[1:])):.3f} ms")
nrm = t.get_normals()
print(os.path.basename(os.environ.get("PCR_LIB", "libpcr_hip.so")), f"n={n}", " ".join(out), "checksum", float(np.abs(nrm).sum(dtype=np.float64)), flush=True)
