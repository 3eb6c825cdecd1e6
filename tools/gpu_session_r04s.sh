#!/bin/bash
# depth of the second list set (and its motion threshold) as compile-time variants
cd "$(dirname "$0")/.."
o=gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for v in h2_025 h2_035 h2_045 h2_035m; do
  echo "== plane_b01 $v"
  PCR_LIB=$PWD/build/exp/libpcr_$v.so timeout 600 python tools/reuse_probe.py --config plane_b01 --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
done; done 2>&1 | tee $o/r04s_set2_depth.txt
for v in h2_025 h2_035 h2_045; do
  echo "== icp_b01 $v"
  PCR_LIB=$PWD/build/exp/libpcr_$v.so timeout 600 python tools/reuse_probe.py --config icp_b01 --reps 4 --modes 0 --tol 1e-3 2>&1 | grep "trajectory total" | head -1
  echo "== plane_b01_resampled $v"
  PCR_LIB=$PWD/build/exp/libpcr_$v.so timeout 600 python tools/reuse_probe.py --config plane_b01_resampled --reps 4 --modes 0 --tol 1e-3 2>&1 | grep "trajectory total" | head -1
done 2>&1 | tee -a $o/r04s_set2_depth.txt
