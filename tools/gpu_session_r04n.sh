#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for v in plain mask5 mask6; do
  export PCR_LIB=$PWD/build/exp/libpcr_$v.so
  echo "== plane_b01 $v"
  timeout 600 python tools/reuse_probe.py --config plane_b01 --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
done; done 2>&1 | tee $o/r04n_rowmask_variants.txt
for v in plain mask5; do
  export PCR_LIB=$PWD/build/exp/libpcr_$v.so
  echo "== plane_100m $v"
  timeout 900 python tools/reuse_probe.py --config plane_100m --reps 2 --modes 0 --tol 1e-3 2>&1 | grep "trajectory total" | head -1
done 2>&1 | tee -a $o/r04n_rowmask_variants.txt
