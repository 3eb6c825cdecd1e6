#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out; export TMPDIR=/tmp
for h in 0 0.1 0.25 0.45; do
  echo "== plane_100m PCR_HALO=$h"
  PCR_HALO=$h timeout 900 python tools/reuse_probe.py --config plane_100m --reps 2 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total\|GN iter\|align" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
done 2>&1 | tee $o/r04k_halo_100m.txt
