#!/bin/bash
# round 6, session g: the whole GPU suite on the frozen library; p2p pass times at N = 2 / 4 / 8; k-NN normals on the lidar cloud
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
S=$root/tools/gpu_session.sh
$S r06g tests
(cd $root && timeout 600 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -s -k "p2p" 2>&1 | grep -E "p2p world|passed|failed" > $o/r06_p2p_ranks.txt; cat $o/r06_p2p_ranks.txt)
(cd $root && timeout 300 python - > $o/r06g_lidar_set_target.txt 2>&1 <<'PY'
import time, numpy as np
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import lidar_sweep, street
import point_cloud_registration_amd as pcr
ctx = _capi.get_context(0)
for name, cloud in (("lidar_sweep", lidar_sweep(1_060_000, 0)), ("street", street(1_060_000, 0))):
    for rep in range(3):
        ctx.synchronize(); t0 = time.perf_counter()
        t = _capi.Target.points(ctx, cloud); ctx.synchronize(); t1 = time.perf_counter()
        t.estimate_normals(15, want=False); ctx.synchronize(); t2 = time.perf_counter()
        v = _capi.Target.voxels(ctx, cloud, 1.0, 10); ctx.synchronize(); t3 = time.perf_counter()
        print(name, "rep", rep, "point index %.3f ms, k-NN normals (k=15) %.3f ms, voxel build %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), t.index_info()["heavy"], flush=True)
        t.close(); v.close()
PY
cat $o/r06g_lidar_set_target.txt | grep -v "^/opt")
