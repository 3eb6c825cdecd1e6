#!/usr/bin/env python3
"""Developer probe of the certified-reuse path: walk the Gauss-Newton trajectory of a bench config in order
(the way align() and bench.py do) with reuse off / automatic / forced and print, per pose, the scan's typical
displacement since the previous pass, the search mode chosen, the fraction of points k_certify left to the
search, and the kernel times.

    python tools/reuse_probe.py [--config plane_b01] [--reps 10] [--scan copy|resampled|crop]
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from point_cloud_registration_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="plane_b01")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--scan", default=None)
ap.add_argument("--tol", default="1e-3,1e-6")
ap.add_argument("--modes", default="0,1,2")
ap.add_argument("--tau", type=float, default=0.0)
ap.add_argument("--mu", type=float, default=0.0)
a = ap.parse_args()

kind_name, n_target, n_scan, voxel_size, desc = B.CONFIGS[a.config]
kind = {"icp": _capi.ICP, "plane": _capi.PLANE, "vplane": _capi.VPLANE, "ndt": _capi.NDT}[kind_name]
ctx = _capi.get_context(0)
target = B.make_cloud(n_target, seed=0, config=a.config)
scan, T_true = B.make_scan(a.config, target, n_scan, a.scan, seed=2)
if kind_name in ("icp", "plane"):
    tgt = _capi.Target.points(ctx, target)
    if kind_name == "plane" and "lidar" in a.config:
        from point_cloud_registration_amd.synthetic import lidar_normals
        tgt.set_normals(lidar_normals(target))
    elif kind_name == "plane":
        tgt.estimate_normals(15, compat=n_target <= 2_000_000, want=False)
else:
    tgt = _capi.Target.voxels(ctx, target, voxel_size, 10)
sc = _capi.Scan(ctx, scan)
T_fin, iters, trace = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
traj = [trace[i, :16].reshape(4, 4).copy() for i in range(iters)]
info = tgt.index_info()
print(f"[{a.config}/{a.scan}] {iters} GN iterations, scan {sc.n}, index cell {info['cell']:.3f} halo {info['halo']:.3f} dims {info.get('dims')} occupied {info.get('occupied')}", flush=True)
ctx.set_reuse(None, a.tau, a.mu)
print("reuse settings", ctx.get_reuse(), flush=True)

names = ("certify", "nn", "reduce")
ref = None
for mode in [int(m) for m in a.modes.split(",")]:
    ctx.set_reuse(mode)
    rows = np.zeros((len(traj), 6)); outs = []
    for r in range(a.reps + 1):
        sc2 = _capi.Scan(ctx, scan)                 # a fresh scan per walk: pass 0 has no history, as in align()
        for k, T in enumerate(traj):
            ctx.profile_enable(True); ctx.profile_reset()
            o = _capi.linearize(tgt, sc2, kind, T, 2.0)
            prof = ctx.profile_read(); ctx.profile_enable(False)
            st = sc2.reuse_stats()
            if r == 0:
                outs.append(o)
                continue
            rows[k, 0] += prof["certify"][1]; rows[k, 1] += prof["nn"][1]; rows[k, 2] += prof["reduce"][1]
            rows[k, 3] = st["last_mode"]; rows[k, 4] = st["last_searched"] / max(sc2.n, 1); rows[k, 5] = st["last_motion"]
        sc2.close()
    rows[:, :3] *= 1e3 / a.reps
    if ref is None:
        ref = outs
    same = all(np.array_equal(x, y) for x, y in zip(ref, outs))
    print(f"-- reuse={mode}  (sums bit-identical to the first mode: {same})")
    for k in range(len(traj)):
        print(f"  pose {k:2d} motion {rows[k, 5] * 1e3:9.2f} mm  mode {'FTL'[int(rows[k, 3])]}  searched {rows[k, 4] * 100:6.2f} %  "
              f"certify {rows[k, 0]:7.1f}  nn {rows[k, 1]:8.1f}  reduce {rows[k, 2]:6.1f}  total {rows[k, :3].sum():8.1f} us", flush=True)
    print(f"  trajectory total {rows[:, :3].sum():9.1f} us  (nn+certify {rows[:, :2].sum():9.1f})", flush=True)
# whole align() calls, host-driven loop (the one certified reuse runs in), reference tolerance and a tight one
for tol in [float(t) for t in a.tol.split(",")]:
    for mode in [int(m) for m in a.modes.split(",")]:
        ctx.set_reuse(mode)
        ts = []
        for r in range(max(a.reps // 2, 3)):
            t0 = time.perf_counter()
            T, it = _capi.align(tgt, sc, kind, np.eye(4), 60, tol, 2.0, _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_HOST_LOOP)
            ts.append(time.perf_counter() - t0)
        st = sc.reuse_stats()
        print(f"align tol={tol:g} reuse={mode}: {it} iterations, median {np.median(ts[1:]) * 1e3:.3f} ms; "
              f"passes F/T/L so far {st['passes_full']}/{st['passes_track']}/{st['passes_list']}, "
              f"list passes searched {100.0 * st['list_searched'] / max(st['list_points'], 1):.1f} % of their points; T[:3,3]={T[:3, 3]}", flush=True)
ctx.set_reuse(1)
