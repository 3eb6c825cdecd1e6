#!/bin/bash
# round 6, session q: heavy targets sorted inside their cells by an 18-bit sub-cell code (64-bit sort key) instead of 6-9 bits
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
(cd $root && timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x -k "lidar or g11 or heavy_index_fuzz" 2>&1 | tail -4 | tee $o/r06q_tests.txt)
for sb in 6 12 18 21; do
  echo "== PCR_HEAVY_SUB_BITS=$sb" | tee -a $o/r06q_poses.txt
  PCR_HEAVY_SUB_BITS=$sb REPS=3 timeout 600 python tools/reuse_probe.py --config plane_lidar --reps 3 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory\|align tol=0.001" | tee -a $o/r06q_poses.txt
done
for sb in 6 18; do
  echo "== PCR_HEAVY_SUB_BITS=$sb" | tee -a $o/r06q_counters.txt
  (PCR_HEAVY_SUB_BITS=$sb PCR_LIB=$root/point_cloud_registration_amd/libpcr_hip_dev.so timeout 600 python tools/lb_counters_probe.py plane_lidar 2>&1 | grep -v "^/opt" | tee -a $o/r06q_counters.txt)
done
python - <<'PY' | tee $o/r06q_set_target.txt
import time, numpy as np, os
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import lidar_sweep
ctx = _capi.get_context(0)
cloud = lidar_sweep(1_060_000, 0)
for rep in range(4):
    ctx.synchronize(); t0 = time.perf_counter()
    t = _capi.Target.points(ctx, cloud); ctx.synchronize(); t1 = time.perf_counter()
    print("lidar_sweep point index %.3f ms" % ((t1 - t0) * 1e3), flush=True)
    t.close()
PY
