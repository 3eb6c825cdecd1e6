import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import bench as B
from point_cloud_registration_amd import _capi
ctx = _capi.get_context(0)
for cfg in sys.argv[1:]:
    kind_name, n_target, n_scan, vs, _ = B.CONFIGS[cfg]
    target = B.make_cloud(n_target, 0); scan, _ = B.make_scan(cfg, target, n_scan)
    tgt = _capi.Target.points(ctx, target); tgt.estimate_normals(15, compat=n_target <= 2_000_000, want=False)
    sc = _capi.Scan(ctx, scan)
    T, it, tr = _capi.align(tgt, sc, _capi.PLANE, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
    for k in sorted(set([0, 1, 2, it // 2, it - 1])):
        c = _capi.nn_counters(tgt, sc, tr[k, :16].reshape(4, 4), 2.0)
        print(cfg, "pose", k, {a: round(b, 2) for a, b in c.items()}, flush=True)
