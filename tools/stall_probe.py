#!/usr/bin/env python3
"""Developer probe: per-pass wall times of an unprofiled linearize loop (looking for spin-wait timeouts)."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, harness_scan
ctx = _capi.get_context(0)
target = street(1_060_000, seed=0); scan = harness_scan(target, 100_000, seed=1)
tgt = _capi.Target.points(ctx, target); sc = _capi.Scan(ctx, scan)
T, it, tr = _capi.align(tgt, sc, _capi.ICP, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
traj = [tr[i, :16].reshape(4, 4).copy() for i in range(it)]
mode = sys.argv[1] if len(sys.argv) > 1 else "prof"
if mode == "prof":
    for r in range(3):
        for k, Tk in enumerate(traj):
            ctx.profile_enable(True); ctx.profile_reset()
            _capi.linearize(tgt, sc, _capi.ICP, Tk, 2.0)
            ctx.profile_read(); ctx.profile_enable(False)
gc.collect(); gc.disable()
ts = []
for k in range(int(os.environ.get("PROBE_PASSES", "60"))):
    t0 = time.perf_counter(); _capi.linearize(tgt, sc, _capi.ICP, traj[k % len(traj)], 2.0); ts.append((time.perf_counter() - t0) * 1e6)
print(mode, "passes", len(ts), "median us", round(float(np.median(ts)), 1), "slow (>500 us):", [(i, round(t)) for i, t in enumerate(ts) if t > 500])
# second part: every pass bracketed by HIP events -> is a slow pass slow on the GPU or on the host side?
gc.enable(); gc.collect(); gc.disable()
slow = []
for k in range(400):
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter(); _capi.linearize(tgt, sc, _capi.ICP, traj[k % len(traj)], 2.0); w = (time.perf_counter() - t0) * 1e6
    prof = ctx.profile_read(); ctx.profile_enable(False)
    kern = sum(v[1] for v in prof.values()) * 1e3
    if w > 500:
        slow.append((k, round(w), round(kern, 1)))
print("slow passes (index, wall us, kernel us by HIP events):", slow)
