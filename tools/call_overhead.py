#!/usr/bin/env python3
"""Fixed cost of one pcr_linearize call (tiny scan): host + 3 launches + completion hand-off."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street
import gc
target = street(200_000, seed=0)
ctx = _capi.get_context(0)
nrm = np.zeros_like(target); nrm[:, 2] = 1
tgt = _capi.Target.points(ctx, target, nrm)
sc = _capi.Scan(ctx, target[:64].copy())
T = np.eye(4)
gc.collect(); gc.disable()
for name, fn in (("python wrapper", lambda: _capi.linearize(tgt, sc, 1, T, 2.0)),):
    for _ in range(200): fn()
    t0 = time.perf_counter()
    for _ in range(2000): fn()
    print(f"{name}: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per call")
# raw ctypes call with preconverted arguments
L = _capi.lib()
out = np.zeros(29); T16 = np.ascontiguousarray(T).reshape(16)
f = L.pcr_linearize
for _ in range(200): f(tgt.handle, sc.handle, 1, T16, 2.0, 1, out)
t0 = time.perf_counter()
for _ in range(2000): f(tgt.handle, sc.handle, 1, T16, 2.0, 1, out)
print(f"raw ctypes (ndpointer argtypes): {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per call")
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(500): f(tgt.handle, sc.handle, 1, T16, 2.0, 1, out)
print({k: round(v[1] / max(v[0], 1) * 1e3, 2) for k, v in ctx.profile_read().items() if v[0]}, "us per kernel (events)")
