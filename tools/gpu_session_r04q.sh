#!/bin/bash
# centroid-grid cell (voxels per cell) re-swept with the filter index at 0.4-cell lists
cd "$(dirname "$0")/.."
o=gpurun_out; export TMPDIR=/tmp
for c in vplane_10m ndt_10m; do
for m in 1.0 1.5 2.0 2.5 3.0; do
  echo "== $c PCR_VOXEL_CELL_MULT=$m"
  PCR_VOXEL_CELL_MULT=$m timeout 600 python tools/reuse_probe.py --config $c --reps 4 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s/%s ", $14, $16; else print }'
done; done 2>&1 | tee $o/r04q_cell_mult.txt
