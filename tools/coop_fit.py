#!/usr/bin/env python3
"""CPU estimate for the wave-cooperative search: for every tile of 64 Morton-consecutive queries, the
box spanned by the lanes' search balls (radius = exact NN distance, the best case of a seeded bound),
the rows some lane needs, and the candidates the wave would test (all points of the needed rows over
the box's x-range).  Compare with the per-lane search's ~14 (converged) .. 84 (first pose) candidates
per lane x ~2.5 divergence factor."""
import numpy as np, sys
sys.path.insert(0, '/root/repo')
from scipy.spatial import cKDTree
from point_cloud_registration_amd.synthetic import street, perturbed_scan, make_T, T_TRUE_SO3, T_TRUE_T
from tools.box_fit import spread

def morton_sort(scan):
    lo = scan.min(0); ext = (scan.max(0) - lo).max(); sc = 2097151.0 / ext
    q = np.clip((scan - lo) * sc, 0, 2097151).astype(np.uint64)
    key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return scan[np.argsort(key, kind='stable')]

def analyse(target, scan, T, h, name, nt=4000, inflate=1.0):
    s = morton_sort(scan)
    st = (T[:3, :3] @ s.T).T + T[:3, 3]
    tree = cKDTree(target)
    tlo = target.min(0)
    dims = (np.floor((target.max(0) - tlo) / h) + 2).astype(int)
    tc = np.floor((target - tlo) / h).astype(int)
    cnt = np.zeros(dims[::-1], np.int64)
    np.add.at(cnt, (tc[:, 2], tc[:, 1], tc[:, 0]), 1)
    rng = np.random.default_rng(0)
    ntiles = len(st) // 64
    pick = rng.choice(ntiles, min(nt, ntiles), replace=False)
    out = []
    for t in pick:
        q = st[t * 64:(t + 1) * 64]
        d, _ = tree.query(q)
        d = np.minimum(d * inflate, 2.0)
        lo = np.clip(np.floor((q - d[:, None] - tlo) / h).astype(int), 0, dims - 1)
        hi = np.clip(np.floor((q + d[:, None] - tlo) / h).astype(int), 0, dims - 1)
        B0 = lo.min(0); B1 = hi.max(0)
        bx, by, bz = (B1 - B0 + 1)
        rows = by * bz
        box = cnt[B0[2]:B1[2] + 1, B0[1]:B1[1] + 1, B0[0]:B1[0] + 1]
        total = box.sum()
        # rows some lane needs: slab distance of the lane to the row <= d
        ys = (np.arange(B0[1], B1[1] + 1))[None, :]; zs = (np.arange(B0[2], B1[2] + 1))[None, :]
        qy = (q[:, 1] - tlo[1])[:, None]; qz = (q[:, 2] - tlo[2])[:, None]
        dy = np.maximum(np.maximum(ys * h - qy, qy - (ys + 1) * h), 0)     # (64, by)
        dz = np.maximum(np.maximum(zs * h - qz, qz - (zs + 1) * h), 0)     # (64, bz)
        need = (dy[:, None, :] ** 2 + dz[:, :, None] ** 2) <= (d ** 2)[:, None, None]   # (64, bz, by)
        rowneed = need.any(0)
        cand_rows = (box.sum(2) * rowneed).sum()
        nonempty_needed = ((box.sum(2) > 0) & rowneed).sum()
        # per-lane candidate count of the per-lane search (all points of cells intersecting the ball's box)
        per_lane = 0
        out.append((bx, by, bz, rows, total, cand_rows, rowneed.sum(), nonempty_needed, d.mean()))
    o = np.array(out, float)
    names = ["bx", "by", "bz", "rows", "box_pts", "cand(needed rows)", "rows needed", "non-empty needed", "mean d"]
    print(name)
    for k, nm in enumerate(names):
        print(f"   {nm:>20}: median {np.median(o[:, k]):8.1f}  mean {o[:, k].mean():8.1f}  p90 {np.percentile(o[:, k], 90):8.1f}  p99 {np.percentile(o[:, k], 99):8.1f}")
    for cap in (256, 384, 512, 768, 1024):
        print(f"   box_pts <= {cap}: {np.mean(o[:, 4] <= cap) * 100:.1f}%   rows<=64: {np.mean(o[:,3]<=64)*100:.1f}%  both: {np.mean((o[:, 4] <= cap)&(o[:,3]<=64)) * 100:.1f}%")

target = street(1_060_000, seed=0)
scan, T_true = perturbed_scan(target, None)
h = 0.405
analyse(target, scan, T_true, h, "b01 converged pose (radius = true NN distance)")
analyse(target, scan, T_true, h, "b01 converged pose, seed radius 2x true", inflate=2.0)
analyse(target, scan, np.eye(4), h, "b01 first pose (radius = true NN distance)")
analyse(target, scan, np.eye(4), h, "b01 first pose, seed radius 1.5x true", inflate=1.5)
