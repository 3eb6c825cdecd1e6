#!/bin/bash
# per-POSE search time against the grid cell (the round-3 sweep recorded trajectory means only): is a coarser grid better at the far poses?
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
for cell in auto 0.3 0.5 0.6 0.7 0.8 1.0 1.2; do
  [ $cell = auto ] && unset PCR_GRID_CELL || export PCR_GRID_CELL=$cell
  for cfg in plane_b01 plane_b01_resampled; do
  echo "== cell $cell $cfg: nn us per pose"
  timeout 600 python tools/reuse_probe.py --config $cfg --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
  done
done 2>&1 | tee $out/r05c_cell_per_pose.txt
