#!/bin/bash
# round 6, session o: p2p exchange timing at N = 2 / 4 / 8 (processes on one GPU), set_target on the lidar cloud, group bench lines, smoke
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
(cd $root && timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_group.py -m gpu -q -s -rs 2>&1 | grep -E "p2p world|group|passed|failed|SKIP" > $o/r06_p2p_ranks.txt; cat $o/r06_p2p_ranks.txt)
(cd $root && timeout 300 python - > $o/r06_lidar_set_target.txt 2>&1 <<'PY'
import time, numpy as np
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import lidar_sweep, street
ctx = _capi.get_context(0)
for name, cloud in (("lidar_sweep", lidar_sweep(1_060_000, 0)), ("street", street(1_060_000, 0))):
    for rep in range(3):
        ctx.synchronize(); t0 = time.perf_counter()
        t = _capi.Target.points(ctx, cloud); ctx.synchronize(); t1 = time.perf_counter()
        t.estimate_normals(15, want=False); ctx.synchronize(); t2 = time.perf_counter()
        v = _capi.Target.voxels(ctx, cloud, 1.0, 10); ctx.synchronize(); t3 = time.perf_counter()
        print(name, "rep", rep, "point index %.3f ms, k-NN normals (k=15) %.3f ms, voxel build %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), "heavy", t.index_info()["heavy"], flush=True)
        t.close(); v.close()
PY
grep -v "^/opt" $o/r06_lidar_set_target.txt)
for n in 4 8; do
  PCR_BENCH_GROUP_DEVICES=$(python -c "print(','.join(['0']*$n))") GPU_MAX_HW_QUEUES=$((n+4)) timeout 600 python bench.py --gpus $n --single-process --no-pmc --no-cpu-baseline --config plane_b01 --steps 20 --warmup 5 > $o/r06_bench_plane_b01_group${n}_1gpu.json 2> $o/r06o_group$n.err; cut -c1-300 $o/r06_bench_plane_b01_group${n}_1gpu.json
done
(cd $root && timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $o/r06_smoke.txt)
(cd $root && timeout 1200 python -m pytest tests/test_gpu_heavy_index.py -m gpu -q --durations=3 2>&1 | tail -6 | tee $o/r06o_heavy.txt)
