#!/usr/bin/env python3
"""Developer probe: fused kernel (variant 0) vs search + reduce (variant 1) per pass along a Gauss-Newton
trajectory, for several scan sizes against the 1.06 M-point B-01 stand-in -> where the automatic choice
(variant 2) should switch."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, perturbed_scan
ctx = _capi.get_context(0)
target = street(1_060_000, seed=0)
for kind_name, kind in (("plane", _capi.PLANE), ("icp", _capi.ICP)):
    tgt = _capi.Target.points(ctx, target)
    if kind == _capi.PLANE:
        tgt.estimate_normals(15, want=False)
    for n in (100_000, 200_000, 262_000, 300_000, 350_000, 400_000, 450_000, 600_000):
        scan, _ = perturbed_scan(target, n if n < 1_060_000 else None, seed=2)
        sc = _capi.Scan(ctx, scan)
        ctx.set_variant(1)
        T, it, tr = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
        traj = [tr[i, :16].reshape(4, 4).copy() for i in range(it)]
        res = {}
        for v in (0, 1):
            ctx.set_variant(v)
            for k in range(10):
                _capi.linearize(tgt, sc, kind, traj[k % len(traj)], 2.0)
            gc.collect(); gc.disable()
            reps = 40 * len(traj)
            t0 = time.perf_counter()
            for k in range(reps):
                _capi.linearize(tgt, sc, kind, traj[k % len(traj)], 2.0)
            res[v] = (time.perf_counter() - t0) / reps * 1e6
            gc.enable()
        ctx.set_variant(2)
        print(f"{kind_name} scan {n:>8}: fused {res[0]:7.1f} us/pass   split {res[1]:7.1f} us/pass   -> {'fused' if res[0] < res[1] else 'split'}", flush=True)
        sc.close()
    tgt.close()
