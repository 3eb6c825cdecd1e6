#!/usr/bin/env python3
"""Soak test of the pass pipeline: random scan sizes / kinds / poses for a fixed wall time; every pass
is evaluated twice and must be bit-identical (catches lost tickets, stale partials, counter races)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, perturbed_scan, make_T

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(0)
target = street(400_000, seed=3)
ctx = _capi.get_context(0)
tp = _capi.Target.points(ctx, target); tp.estimate_normals(10, want=False)
tv = _capi.Target.voxels(ctx, target, 1.0, 10)
full, _ = perturbed_scan(target, None, seed=4)
t0 = time.time(); passes = 0; scans = 0
rebuilds = 0
while time.time() - t0 < seconds:
    if scans % 8 == 7:                          # targets come and go too: their blocks are recycled through the context's cache
        m = int(rng.choice([50_000, 200_000, 400_000, int(rng.integers(20_000, 400_000))]))
        sub = target[np.sort(rng.permutation(len(target))[:m])].copy()
        tp.close(); tv.close()
        tp = _capi.Target.points(ctx, sub); tp.estimate_normals(10, want=False)
        tv = _capi.Target.voxels(ctx, sub, 1.0, 10)
        q = full[rng.integers(0, len(full), 256)]
        d, i = tp.nn_query(q)
        db = np.sqrt(((q[:, None, :].astype(np.float64) - sub[None, :, :].astype(np.float64)) ** 2).sum(-1).min(1))
        if not np.allclose(d, db, rtol=1e-5, atol=1e-6):
            print("TARGET MISMATCH after rebuild", m, np.max(np.abs(d - db))); sys.exit(1)
        rebuilds += 1
    n = int(rng.choice([1, 7, 63, 64, 65, 500, 2047, 2048, 2049, 30_000, 131_072, 131_073, 250_000, int(rng.integers(1, 400_000))]))
    sc = _capi.Scan(ctx, full[rng.permutation(len(full))[:n]].copy()); scans += 1
    for _ in range(40):
        kind = int(rng.integers(0, 4))
        T = make_T(rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3))
        tgt = tp if kind < 2 else tv
        a = _capi.linearize(tgt, sc, kind, T, 2.0)
        b = _capi.linearize(tgt, sc, kind, T, 2.0)
        if not np.array_equal(a, b) or not np.isfinite(a).all() or a[28] > n:
            print("MISMATCH", n, kind, a[:3], b[:3], a[28], b[28]); sys.exit(1)
        passes += 2
        if passes % 20 == 0:                    # the other kernel pipelines: same correspondences, same sums up to summation order
            for pipe in (dict(variant=0), dict(variant=1), dict(variant=1, reuse=2), dict(variant=1, reuse=0)) + ((dict(variant=1, nn_mode=2), dict(variant=1, fuse_finalize=0)) if _capi.has_dev_kernels() else ()):
                with ctx.pipeline(**pipe):
                    c = _capi.linearize(tgt, sc, kind, T, 2.0)
                if c[28] != a[28] or not np.allclose(c, a, rtol=1e-10, atol=1e-9 * max(np.max(np.abs(a)), 1.0)):
                    print("PIPELINE MISMATCH", pipe, n, kind, a[:3], c[:3], a[28], c[28]); sys.exit(1)
                passes += 1
        if passes % 50 == 0 and n > 100:        # device-resident loop == host-driven loop, bit for bit
            try:
                Td, itd = _capi.align(tgt, sc, kind, T, 10, 1e-3, 2.0, _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_DEVICE_LOOP)
                Th, ith = _capi.align(tgt, sc, kind, T, 10, 1e-3, 2.0, _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_HOST_LOOP)
            except np.linalg.LinAlgError:
                Th, ith = None, None
            if Th is not None and (itd != ith or not np.array_equal(Td, Th)):
                print("LOOP MISMATCH", n, kind, itd, ith, "index", tgt.index_info())
                # diagnostics: the two loops again with their traces -- where do the sums part?
                Td2, itd2, trd = _capi.align(tgt, sc, kind, T, 10, 1e-3, 2.0, _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_DEVICE_LOOP, want_trace=True)
                Th2, ith2, trh = _capi.align(tgt, sc, kind, T, 10, 1e-3, 2.0, _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_HOST_LOOP, want_trace=True)
                print("  again: equal now?", np.array_equal(Td2, Th2), "device run reproducible?", np.array_equal(Td, Td2), "host run reproducible?", np.array_equal(Th, Th2))
                for k in range(min(itd2, ith2)):
                    if not np.array_equal(trd[k], trh[k]):
                        print("  first differing iteration", k, "pose equal", np.array_equal(trd[k, :16], trh[k, :16]), "count", trd[k, 44], trh[k, 44],
                              "max rel diff of the sums", np.max(np.abs(trd[k, 16:] - trh[k, 16:])) / np.max(np.abs(trh[k, 16:])))
                        break
                sys.exit(1)
    sc.close()
print(f"soak ok: {passes} passes over {scans} scans, {rebuilds} target rebuilds in {time.time() - t0:.1f} s")
