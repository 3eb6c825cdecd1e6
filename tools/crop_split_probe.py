#!/usr/bin/env python3
"""Developer probe (round 5): where do the ~200 us per search of bench config plane_b01_crop go?  The scan's points that land
inside / outside the map's bounding box (at the converged pose), searched separately."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
import bench
ctx = _capi.get_context(0)
target = bench.make_cloud(1_060_000, seed=0)
scan, T_true = bench.make_scan("plane_b01_crop", target, 1_060_000, seed=2)
tgt = _capi.Target.points(ctx, target); tgt.estimate_normals(15, want=False)
st = scan.astype(np.float64) @ T_true[:3, :3].T + T_true[:3, 3]
lo, hi = target.min(0), target.max(0)
inside = np.all((st >= lo - 2.0) & (st <= hi + 2.0), axis=1)
print("points inside the map's box (+2 m):", int(inside.sum()), "of", scan.shape[0])
def timed(pts, tag):
    sc = _capi.Scan(ctx, np.ascontiguousarray(pts))
    for P, name in ((np.eye(4), "identity"), (T_true, "T_true")):
        for _ in range(4):
            _capi.linearize(tgt, sc, _capi.PLANE, P, 2.0)
        ctx.profile_enable(True); ctx.profile_reset()
        for _ in range(10):
            out = _capi.linearize(tgt, sc, _capi.PLANE, P, 2.0)
        prof = ctx.profile_read(); ctx.profile_enable(False)
        print(f"{tag:<22} {pts.shape[0]:8d} points at {name:<9} " + ", ".join(f"{k} {v[1] / v[0] * 1e3:.1f} us" for k, v in prof.items() if v[0]) + f"; correspondences {int(out[28])}")
    sc.close()
timed(scan, "whole crop scan")
timed(scan[inside], "inside only")
timed(scan[~inside], "outside only")

if _capi.has_dev_kernels():
    for pts, tag in ((scan[inside], "inside"), (scan[~inside], "outside")):
        sc = _capi.Scan(ctx, np.ascontiguousarray(pts))
        print(tag, "work counters per query at T_true:", {k: round(v, 2) for k, v in _capi.nn_counters(tgt, sc, T_true, 2.0).items()})
        sc.close()
    # the outside points by how far outside the grid's box they are (cells)
    info = tgt.index_info()
    h = info["cell"]
    d = np.maximum(np.maximum(lo - st, st - hi), 0.0).max(1) / h
    out = scan[~inside]; do = d[~inside]
    for a, b in ((0, 6), (6, 8), (8, 16), (16, 64), (64, 1e9)):
        sel = (do >= a) & (do < b)
        if sel.sum() < 1000:
            print(f"outside by [{a}, {b}) cells: {int(sel.sum())} points"); continue
        sc = _capi.Scan(ctx, np.ascontiguousarray(out[sel]))
        for _ in range(3): _capi.linearize(tgt, sc, _capi.PLANE, T_true, 2.0)
        ctx.profile_enable(True); ctx.profile_reset()
        for _ in range(10): _capi.linearize(tgt, sc, _capi.PLANE, T_true, 2.0)
        prof = ctx.profile_read(); ctx.profile_enable(False)
        nn = prof['nn'] if prof['nn'][0] else prof['linearize']
        print(f"outside by [{a}, {b}) cells: {int(sel.sum())} points, search {nn[1] / max(nn[0], 1) * 1e3:.1f} us; counters", {k: round(v, 2) for k, v in _capi.nn_counters(tgt, sc, T_true, 2.0).items()})
        sc.close()
