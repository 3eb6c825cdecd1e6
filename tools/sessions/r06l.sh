#!/bin/bash
# round 6, session l: joint sweep of the centroid filter index's cell edge (voxels) and list depth (cells) -- VERDICT r5 item 6
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
for cfg in vplane_10m ndt_10m; do
  for mh in "2.0 0.4" "1.0 0.5" "1.0 0.7" "1.0 1.0" "1.5 0.4" "1.5 0.55" "1.5 0.7" "1.25 0.6" "1.25 0.8"; do
    set -- $mh
    echo "== $cfg PCR_VOXEL_CELL_MULT=$1 PCR_FILTER_HALO=$2" | tee -a $o/r06l_filter_sweep.txt
    PCR_VOXEL_CELL_MULT=$1 PCR_FILTER_HALO=$2 timeout 300 python tools/reuse_probe.py --config $cfg --reps 3 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory\|align" | awk '/pose/ {printf "%s/%s ", $(NF-6), $(NF-4)} /trajectory/ {print; } /align/ {print}' | tee -a $o/r06l_filter_sweep.txt
  done
done
for cfg in vplane_10m plane_b01; do
  rm -rf $o/prof_gaps
  (cd /tmp; timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $o/prof_gaps -o r -- python $root/tools/pass_gaps.py run $cfg > $o/prof_gaps.log 2>&1)
  db=$(find $o/prof_gaps -name "*.db" | head -1)
  { echo "== $cfg"; python tools/pass_gaps.py show "$db"; } 2>&1 | tee -a $o/r06l_pass_gaps.txt
  rm -rf $o/prof_gaps
done
(cd $root && timeout 600 python -m pytest tests/test_gpu_phase_split.py -m gpu -x -q 2>&1 | tail -5 | tee $o/r06l_ps_test.txt)
