#!/bin/bash
# filter-index list depth on the voxel configs (PCR_HALO reaches the float32 filter index of a voxel target)
cd "$(dirname "$0")/.."
o=gpurun_out; export TMPDIR=/tmp
for c in vplane_10m ndt_10m; do
for h in 0.5 0.8 1.0; do
  echo "== $c PCR_HALO=$h"
  PCR_HALO=$h timeout 600 python tools/reuse_probe.py --config $c --reps 4 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s/%s ", $14, $16; else print }'
done; done 2>&1 | tee $o/r04o_filter_halo_b.txt
