#!/usr/bin/env python3
"""Developer probe: per-pose kernel times along the Gauss-Newton trajectory of a bench config, for
each NN mode, plus whole-align timings of the device-resident loop vs the host-driven one.

    python tools/pose_profile.py [--config plane_b01] [--reps 20] [--modes 0,1]
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import harness_scan, perturbed_scan

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="plane_b01")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--modes", default="0,1")
ap.add_argument("--align-reps", type=int, default=10)
ap.add_argument("--save-traj", default=None)
ap.add_argument("--load-traj", default=None, help="poses from a correct build (ablation builds give wrong sums)")
ap.add_argument("--brief", action="store_true", help="per-pose sequence timings only")
a = ap.parse_args()

kind_name, n_target, n_scan, voxel_size, desc = B.CONFIGS[a.config]
kind = {"icp": _capi.ICP, "plane": _capi.PLANE, "vplane": _capi.VPLANE, "ndt": _capi.NDT}[kind_name]
ctx = _capi.get_context(0)
t0 = time.time()
target = B.make_cloud(n_target, seed=0)
if "harness" in a.config:
    scan = harness_scan(target, n_scan, seed=1)
else:
    scan, _ = perturbed_scan(target, n_scan if n_scan < n_target else None, seed=2)
print(f"[{a.config}] data {time.time() - t0:.1f}s", flush=True)
if kind_name in ("icp", "plane"):
    tgt = _capi.Target.points(ctx, target)
    if kind_name == "plane":
        tgt.estimate_normals(15, compat=n_target <= 2_000_000, want=False)
else:
    tgt = _capi.Target.voxels(ctx, target, voxel_size, 10)
sc = _capi.Scan(ctx, scan)
if a.load_traj:
    traj = list(np.load(a.load_traj)); iters = len(traj)
else:
    T_fin, iters, trace = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
    traj = [trace[i, :16].reshape(4, 4).copy() for i in range(iters)]
if a.save_traj:
    np.save(a.save_traj, np.array(traj))
print(f"[{a.config}] {iters} GN iterations, index {tgt.index_info()}", flush=True)

for mode in [int(m) for m in a.modes.split(",")]:
    ctx.set_nn_mode(mode)
    # (1) each pose on its own, repeated (seeded mode: seeded by the same pose's matches = best case)
    for k, T in enumerate([] if a.brief else traj):
        _capi.linearize(tgt, sc, kind, T, 2.0)
        ctx.profile_enable(True); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            _capi.linearize(tgt, sc, kind, T, 2.0)
        wall = (time.perf_counter() - t0) / a.reps * 1e3
        prof = ctx.profile_read(); ctx.profile_enable(False)
        ks = " ".join(f"{n}={v[1] / max(v[0], 1) * 1e3:.1f}us" for n, v in prof.items() if v[0])
        print(f"  mode {mode} pose {k} repeated : wall(profiled) {wall * 1e3:.1f} us | {ks}", flush=True)
    # (2) walking the trajectory cyclically like bench.py does (each pass seeded by the previous pose)
    per = np.zeros((len(traj), 2)); cnt = np.zeros(len(traj))
    for r in range(a.reps):
        for k, T in enumerate(traj):
            ctx.profile_enable(True); ctx.profile_reset()
            _capi.linearize(tgt, sc, kind, T, 2.0)
            prof = ctx.profile_read(); ctx.profile_enable(False)
            if r > 0:
                per[k, 0] += prof["nn"][1]; per[k, 1] += prof["reduce"][1]; cnt[k] += 1
    for k in range(len(traj)):
        print(f"  mode {mode} pose {k} in sequence: nn={per[k, 0] / cnt[k] * 1e3:.1f}us reduce={per[k, 1] / cnt[k] * 1e3:.1f}us", flush=True)
    print(f"  mode {mode} trajectory mean: nn={per[:, 0].sum() / cnt.sum() * 1e3:.1f}us reduce={per[:, 1].sum() / cnt.sum() * 1e3:.1f}us", flush=True)
    if a.brief:
        continue
    # (3) unprofiled wall per pass, trajectory walk (what bench.py reports)
    import gc; gc.collect(); gc.disable()
    for k in range(5):
        _capi.linearize(tgt, sc, kind, traj[k % len(traj)], 2.0)
    n = a.reps * len(traj)
    t0 = time.perf_counter()
    for k in range(n):
        _capi.linearize(tgt, sc, kind, traj[k % len(traj)], 2.0)
    print(f"  mode {mode} walk wall {((time.perf_counter() - t0) / n) * 1e6:.1f} us/pass", flush=True)
    # (4) whole align(): device-resident loop vs host-driven loop (scan resident)
    for name, fl in (("device", _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_DEVICE_LOOP), ("host", _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_HOST_LOOP), ("auto", _capi.FLAG_ICP_RR_QUIRK)):
        ts = []
        for r in range(a.align_reps):
            t0 = time.perf_counter()
            T, it = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, fl)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts[1:]) * 1e6
        print(f"  mode {mode} align[{name} loop]: {it} iterations, median {np.median(ts):.1f} us "
              f"(min {ts.min():.1f}) = {np.median(ts) / it:.1f} us/iteration", flush=True)
    gc.enable()
ctx.set_nn_mode(0)
