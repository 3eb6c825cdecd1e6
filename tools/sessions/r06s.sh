#!/bin/bash
# round 6, session s: cell edge of the heavy index under the final scheme (nearest-first, 16-byte boxes, 18-bit sub-cell order)
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
for h in auto 0.32 0.4 0.5 0.64 0.8; do
  echo "== PCR_GRID_CELL=$h" | tee -a $o/r06s_cell.txt
  if [ $h = auto ]; then unset PCR_GRID_CELL; else export PCR_GRID_CELL=$h; fi
  timeout 600 python tools/reuse_probe.py --config plane_lidar --reps 3 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory\|align tol=0.001" | awk '/pose/ {printf "%s ", $(NF-6)} /trajectory/ {print} /align/ {print}' | tee -a $o/r06s_cell.txt
  timeout 300 python bench.py --config icp_lidar_harness --no-pmc --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('icp_lidar_harness', d['value'], d['ms_per_step'], d['kernels'], 'set_target', d['config']['set_target_ms'], 'first align', d['config']['first_align_ms'], d['config']['nn_index']['cell'])" | tee -a $o/r06s_cell.txt
done
