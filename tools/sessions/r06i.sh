#!/bin/bash
# round 6, session i: phase-split search + reduce kernel (k_scan_reduce) -- exactness subset, A/B per pose against the kernel pair, 4 vs 5 waves
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
(cd $root && timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linearize or align_matches or fuzz_align or robustness" > $o/r06i_quick.log 2>&1; echo "rc=$?" >> $o/r06i_quick.log; tail -5 $o/r06i_quick.log)
for cfg in plane_b01 icp_b01; do
  for ps in 0 1; do
    PCR_PHASE_SPLIT=$ps timeout 300 python tools/phase_split_probe.py --config $cfg 2>&1 | grep -v "^/opt" | tee -a $o/r06i_ps_probe.txt
  done
  PCR_LIB=$root/point_cloud_registration_amd/variants/libpcr_hip_ps4.so timeout 300 python tools/phase_split_probe.py --config $cfg 2>&1 | grep -v "^/opt" | tee -a $o/r06i_ps_probe.txt
done
for cfg in plane_b01_resampled plane_b01_crop; do
  for ps in 0 1; do
    PCR_PHASE_SPLIT=$ps timeout 300 python tools/phase_split_probe.py --config $cfg 2>&1 | grep -v "^/opt" | tee -a $o/r06i_ps_probe.txt
  done
done
BENCH_ARGS="--no-pmc --no-cpu-baseline" timeout 400 tools/gpu_session.sh r06i bench:default
