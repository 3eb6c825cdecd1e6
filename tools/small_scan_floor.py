#!/usr/bin/env python3
"""Developer probe: per-pass wall time of the fused small-scan kernel against the 1.06 M-point target as the
scan shrinks -- how much of a 100 k-point pass is the fixed chain (launch, tile hand-out, fold, hand-off to the
host) and how much the search."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, perturbed_scan
ctx = _capi.get_context(0)
target = street(1_060_000, seed=0)
tgt = _capi.Target.points(ctx, target)
tgt.estimate_normals(15, want=False)
full, _ = perturbed_scan(target, None, seed=2)
rng = np.random.default_rng(0)
for kind_name, kind in (("icp", _capi.ICP), ("plane", _capi.PLANE)):
    for n in (64, 1_000, 10_000, 30_000, 100_000, 200_000):
        sc = _capi.Scan(ctx, full[rng.permutation(len(full))[:n]].copy())
        T, it, tr = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
        traj = [tr[i, :16].reshape(4, 4).copy() for i in range(it)]
        out = []
        for poses in (traj[-1:], traj[:1]):
            for k in range(20):
                _capi.linearize(tgt, sc, kind, poses[k % len(poses)], 2.0)
            gc.collect(); gc.disable()
            reps = 400
            t0 = time.perf_counter()
            for k in range(reps):
                _capi.linearize(tgt, sc, kind, poses[k % len(poses)], 2.0)
            out.append((time.perf_counter() - t0) / reps * 1e6)
            gc.enable()
        print(f"{kind_name} scan {n:>7}: converged pose {out[0]:6.1f} us/pass   first pose {out[1]:6.1f} us/pass", flush=True)
        sc.close()
