#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
for h in 0.1 0.2 0.3 0.4; do
  for c in plane_b01 plane_b01_resampled; do
    echo "=== halo $h $c"
    PCR_HALO=$h timeout 600 python tools/reuse_probe.py --config $c --reps 8 --modes 1 --tol 1e-3 2>&1 | grep "pose\|total\|GN iter"
  done
done
