#!/bin/bash
# round 6, session j: phase-split kernel with more than one resident generation of blocks (reduce of early blocks under the search of later ones)
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
for m in 1 2 3 4; do
  echo "== ps4 grid mult $m" | tee -a $o/r06j_ps_probe.txt
  PCR_PS_GRID_MULT=$m PCR_LIB=$root/point_cloud_registration_amd/variants/libpcr_hip_ps4.so timeout 300 python tools/phase_split_probe.py --config plane_b01 2>&1 | grep -v "^/opt" | tee -a $o/r06j_ps_probe.txt
done
for m in 2 3; do
  echo "== ps5 grid mult $m" | tee -a $o/r06j_ps_probe.txt
  PCR_PS_GRID_MULT=$m timeout 300 python tools/phase_split_probe.py --config plane_b01 2>&1 | grep -v "^/opt" | tee -a $o/r06j_ps_probe.txt
done
