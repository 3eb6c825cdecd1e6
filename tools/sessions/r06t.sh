#!/bin/bash
# round 6, session t: wave-aggregated atomics in the probing histogram of the index build (heavy clouds)
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
(cd $root && timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x -k "lidar or g11 or heavy_index_fuzz or nn_query or fuzz_against_oracle or voxel_build or degenerate" 2>&1 | tail -3 | tee $o/r06t_tests.txt)
python - <<'PY' | tee $o/r06t_set_target.txt
import time, numpy as np
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import lidar_sweep, street, street_tiled
ctx = _capi.get_context(0)
for name, cloud in (("lidar_sweep 1.06M", lidar_sweep(1_060_000, 0)), ("street 1.06M", street(1_060_000, 0)), ("street_tiled 10M", street_tiled(10_000_000, 0))):
    ts = []
    for rep in range(6):
        ctx.synchronize(); t0 = time.perf_counter()
        t = _capi.Target.points(ctx, cloud); ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        info = t.index_info(); t.close()
    print(name, "point index ms:", " ".join(f"{v:.3f}" for v in ts), "cell", round(info["cell"], 4), "heavy", info["heavy"], flush=True)
PY
rm -rf $o/prof_tl
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd -d $o/prof_tl -o r -- python $root/tools/build_timeline.py run index 1.06e6 > /dev/null 2>&1)
db=$(find $o/prof_tl -name "*.db" | head -1); python tools/rocpd_summary.py "$db" 2>&1 | grep "k_cell_ids\|k_bbox\|radix" | head -5 | tee -a $o/r06t_set_target.txt; rm -rf $o/prof_tl
