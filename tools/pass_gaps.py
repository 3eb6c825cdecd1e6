#!/usr/bin/env python3
"""Developer probe: where the time of a host-driven pass goes that is not kernel time.  `run` walks a bench config's trajectory
(the bench's own steps, no events); `show` reads the rocprofv3 rocpd database of that run and prints, over the last passes, the mean
kernel durations and the idle gaps search -> reduce and reduce -> next search (host round trip + launch).
    rocprofv3 --kernel-trace --output-format rocpd -d out -o r -- python tools/pass_gaps.py run vplane_10m
    python tools/pass_gaps.py show out/.../r_results.db"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(config):
    import numpy as np
    import bench as B
    from point_cloud_registration_amd import _capi
    kind_name, n_target, n_scan, voxel_size, desc = B.CONFIGS[config]
    kind = {"icp": _capi.ICP, "plane": _capi.PLANE, "vplane": _capi.VPLANE, "ndt": _capi.NDT}[kind_name]
    ctx = _capi.get_context(0)
    target = B.make_cloud(n_target, seed=0, config=config)
    scan, T_true = B.make_scan(config, target, n_scan, None, seed=2)
    if kind_name in ("icp", "plane"):
        tgt = _capi.Target.points(ctx, target)
        if kind_name == "plane":
            tgt.estimate_normals(15, compat=n_target <= 2_000_000, want=False)
    else:
        tgt = _capi.Target.voxels(ctx, target, voxel_size, 10)
    sc = _capi.Scan(ctx, scan)
    T_fin, iters, trace = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
    traj = [trace[i, :16].reshape(4, 4).copy() for i in range(iters)]
    for r in range(8):
        for T in traj:
            _capi.linearize(tgt, sc, kind, T, 2.0)
    print(config, "passes", 8 * len(traj), flush=True)


def show(db):
    import re, sqlite3
    cur = sqlite3.connect(db).cursor()
    ev = [(s, e, re.sub(r"\(.*", "", nm).replace("void ", "")) for nm, s, e in cur.execute("select name, start, end from kernels").fetchall()]
    ev.sort()
    hot = [(s, e, nm) for s, e, nm in ev if re.match(r"k_nn_|k_reduce_|k_linearize|k_scan_reduce", nm)]
    hot = hot[-60:]
    gaps = {}
    durs = {}
    for (s0, e0, n0), (s1, e1, n1) in zip(hot[:-1], hot[1:]):
        gaps.setdefault(n0[:28] + " -> " + n1[:28], []).append((s1 - e0) / 1e3)
    for s, e, nm in hot:
        durs.setdefault(nm[:40], []).append((e - s) / 1e3)
    for k, v in durs.items():
        print(f"kernel {k:42s} n {len(v):3d}  mean {sum(v) / len(v):9.1f} us")
    for k, v in gaps.items():
        v2 = sorted(v)
        print(f"gap    {k:60s} n {len(v):3d}  median {v2[len(v2) // 2]:7.1f}  mean {sum(v) / len(v):7.1f}  min {v2[0]:7.1f} us")


if __name__ == "__main__":
    run(sys.argv[2]) if sys.argv[1] == "run" else show(sys.argv[2])
