#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
for v in "auto:" "forced local:PCR_TILE_LOCAL=1"; do
  name=${v%%:*}; e=${v#*:}
  for cfg in vplane_10m ndt_10m plane_b01_crop plane_100m; do
  echo "== $name $cfg"
  env $e timeout 900 python tools/reuse_probe.py --config $cfg --reps $([ $cfg = plane_100m ] && echo 2 || echo 5) --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
  done
done 2>&1 | tee $out/r05p_handout_policy_large.txt
