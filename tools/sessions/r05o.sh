#!/bin/bash
# hand-out policy with the interleave on: forced global counters / block-local, and the step size below which the hand-out turns block-local
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
for v in "auto:" "forced global:PCR_TILE_LOCAL=0" "forced local:PCR_TILE_LOCAL=1" "frac 0.15:PCR_LOCAL_FRAC=0.15" "frac 0.7:PCR_LOCAL_FRAC=0.7" "frac 1.5:PCR_LOCAL_FRAC=1.5"; do
  name=${v%%:*}; e=${v#*:}
  for cfg in plane_b01 icp_b01 plane_b01_resampled; do
  echo "== $name $cfg"
  env $e timeout 600 python tools/reuse_probe.py --config $cfg --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
  done
done 2>&1 | tee $out/r05o_handout_policy.txt
