#!/usr/bin/env python3
"""Developer timing probe (not the contract bench): per-kernel ms for each kind / variant / cell size."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, street_tiled, perturbed_scan, harness_scan

ap = argparse.ArgumentParser()
ap.add_argument("--nt", type=float, default=1.06e6)
ap.add_argument("--ns", type=float, default=1.06e6)
ap.add_argument("--cells", type=str, default="0")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--mode", default="perturbed")
ap.add_argument("--kinds", default="plane")
a = ap.parse_args()
nt, ns = int(a.nt), int(a.ns)
t0 = time.time()
target = street(nt, seed=0) if nt <= 2_000_000 else street_tiled(nt, seed=0)
if a.mode == "perturbed":
    scan, T_true = perturbed_scan(target, ns if ns < nt else None)
else:
    scan = harness_scan(target, ns)
rng = np.random.default_rng(0)
normals = np.zeros_like(target); normals[:, 2] = 1
print(f"gen {time.time()-t0:.1f}s nt={nt} ns={scan.shape[0]}", flush=True)
ctx = _capi.get_context(0)
T = np.eye(4)
for cell in [float(c) for c in a.cells.split(",")]:
    t0 = time.time()
    tgt = _capi.Target.points(ctx, target, normals, cell_hint=cell)
    t1 = time.time()
    sc = _capi.Scan(ctx, scan)
    t2 = time.time()
    info = tgt.index_info()
    print(f"cell_hint={cell} -> cell={info['cell']:.4f} dims={info['dims']} occupied={info['occupied']} "
          f"occ={info['n']/max(info['occupied'],1):.2f} build={t1-t0:.3f}s scan_upload={t2-t1:.3f}s", flush=True)
    for kind_name in a.kinds.split(","):
        kind = {"icp": 0, "plane": 1}[kind_name]
        for variant in (0, 1):
            ctx.set_variant(variant)
            _capi.linearize(tgt, sc, kind, T, 2.0)
            ctx.profile_enable(True); ctx.profile_reset()
            t0 = time.time()
            for _ in range(a.iters):
                out = _capi.linearize(tgt, sc, kind, T, 2.0)
            wall = (time.time() - t0) / a.iters * 1e3
            prof = ctx.profile_read(); ctx.profile_enable(False)
            ks = " ".join(f"{k}={v[1]/max(v[0],1):.3f}ms" for k, v in prof.items() if v[0])
            print(f"  {kind_name} variant={variant}: wall {wall:.3f} ms/iter  corr={int(out[28])}  "
                  f"{scan.shape[0]/wall/1e3:.1f} Mcorr/s | {ks}", flush=True)
    ctx.set_variant(1)
    tgt.close(); sc.close()
