#!/usr/bin/env python3
"""Developer probe of the phase-split search + reduce kernel (k_scan_reduce, round 6): per pose of a bench config's
Gauss-Newton trajectory the kernel times (HIP events of the library) and the wall time of a pass, then whole align() calls
(device-resident loop).  Run once per setting: PCR_PHASE_SPLIT=0 / 1, PCR_LIB=<variant>.

    python tools/phase_split_probe.py [--config plane_b01] [--reps 20]
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from point_cloud_registration_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="plane_b01")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--scan", default=None)
a = ap.parse_args()
kind_name, n_target, n_scan, voxel_size, desc = B.CONFIGS[a.config]
kind = {"icp": _capi.ICP, "plane": _capi.PLANE, "vplane": _capi.VPLANE, "ndt": _capi.NDT}[kind_name]
ctx = _capi.get_context(0)
target = B.make_cloud(n_target, seed=0, config=a.config)
scan, T_true = B.make_scan(a.config, target, n_scan, a.scan, seed=2)
tgt = _capi.Target.points(ctx, target)
if kind_name == "plane":
    tgt.estimate_normals(15, compat=n_target <= 2_000_000, want=False)
sc = _capi.Scan(ctx, scan)
T_fin, iters, trace = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
traj = [trace[i, :16].reshape(4, 4).copy() for i in range(iters)]
print(f"[{a.config}] PCR_PHASE_SPLIT={os.environ.get('PCR_PHASE_SPLIT', '(default)')} PCR_LIB={os.environ.get('PCR_LIB', '(shipped)')}: {iters} iterations", flush=True)
# warm the lazily built second list set (12 passes) the way bench.py's steady state sees it
for r in range(4):
    for T in traj:
        _capi.linearize(tgt, sc, kind, T, 2.0)
tot_k = 0.0; tot_w = 0.0
for k, T in enumerate(traj):
    ctx.profile_enable(True); ctx.profile_reset()
    for r in range(a.reps):
        o = _capi.linearize(tgt, sc, kind, T, 2.0)
    prof = ctx.profile_read(); ctx.profile_enable(False)
    t0 = time.perf_counter()
    for r in range(a.reps):
        o = _capi.linearize(tgt, sc, kind, T, 2.0)
    w = (time.perf_counter() - t0) / a.reps * 1e6
    ks = {n: prof[n][1] / max(prof[n][0], 1) * 1e3 for n in ("linearize", "nn", "reduce")}
    tot_k += sum(ks.values()); tot_w += w
    print(f"  pose {k}: linearize {ks['linearize']:7.1f}  nn {ks['nn']:7.1f}  reduce {ks['reduce']:6.1f}  kernels {sum(ks.values()):7.1f} us   wall/pass {w:7.1f} us   e2 {o[27]:.12e} H00 {o[0]:.12e}", flush=True)
print(f"  trajectory: kernels {tot_k:8.1f} us, wall {tot_w:8.1f} us  ({sc.n * len(traj) / tot_w:.0f} M corr/s by the wall)", flush=True)
ts = []
for r in range(12):
    t0 = time.perf_counter()
    T, it = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0)
    ts.append(time.perf_counter() - t0)
print(f"  align (device-resident loop): {it} iterations, median {np.median(ts[2:]) * 1e3:.4f} ms, min {np.min(ts[2:]) * 1e3:.4f} ms; T[:3,3] = {T[:3, 3]}", flush=True)
