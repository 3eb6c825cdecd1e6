"""Developer probe (libpcr_hip_dev.so): work counters of the per-lane search on a bench config, per pose -- record / box LOADS per
query (`candidates`; boxes count 2 each in a heavy-cell index), rows, rings, and the same with each wave's slowest lane charged
to all 64.   PCR_LIB=point_cloud_registration_amd/libpcr_hip_dev.so python tools/lb_counters_probe.py plane_lidar [plane_b01]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from point_cloud_registration_amd import _capi
ctx = _capi.get_context(0)
for cfg in sys.argv[1:]:
    kind_name, n_target, n_scan, vs, _ = B.CONFIGS[cfg]
    target = B.make_cloud(n_target, 0, cfg); scan, _ = B.make_scan(cfg, target, n_scan)
    tgt = _capi.Target.points(ctx, target)
    if "lidar" in cfg:
        from point_cloud_registration_amd.synthetic import lidar_normals
        tgt.set_normals(lidar_normals(target))
    else:
        tgt.estimate_normals(15, compat=n_target <= 2_000_000, want=False)
    sc = _capi.Scan(ctx, scan)
    print(cfg, tgt.index_info(), flush=True)
    T, it, tr = _capi.align(tgt, sc, _capi.PLANE, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
    for k in sorted(set([0, 1, 2, it // 2, it - 1])):
        c = _capi.nn_counters(tgt, sc, tr[k, :16].reshape(4, 4), 2.0)
        print(cfg, "pose", k, {a: round(b, 2) for a, b in c.items()}, flush=True)
