#!/bin/bash
# counterfactual: what the search would cost if a candidate batch were ONE load instruction (results wrong by construction)
cd "$(dirname "$0")/.."
o=gpurun_out; export TMPDIR=/tmp
for v in plain ablate_loads; do
  echo "== plane_b01 $v"
  PCR_LIB=$PWD/build/exp/libpcr_$v.so timeout 600 python tools/reuse_probe.py --config plane_b01 --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
done 2>&1 | tee $o/r04l_ablate_loads.txt
