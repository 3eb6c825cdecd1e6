#!/bin/bash
# Closing session, part B (rocprofv3 kernel-trace / PMC summaries, set_target side, timelines, parity margins): GPU suite (ship + developer build), every bench config with live PMC traffic,
# usage: tools/gpu_session_final.sh <tag>   (tag = r05 ...: every output lands in gpurun_out/<tag>_*; copy what should be judged into profiles/)
# rocprofv3 kernel-trace / PMC summaries, per-pose probes, set_target side, seam probes, soak, rare-event trace, 2-rank bench.
cd "$(dirname "$0")/.."; TAG=${1:-r06}
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
export PCR_BENCH_NO_PMC=1
tools/collect_profiles.sh ${TAG}_plane_b01 plane_b01
tools/collect_profiles.sh ${TAG}_vplane_10m vplane_10m
tools/collect_profiles.sh ${TAG}_ndt_10m ndt_10m
tools/collect_profiles.sh ${TAG}_icp_b01_harness icp_b01_harness
tools/collect_profiles.sh ${TAG}_vplane_b01_harness vplane_b01_harness
tools/collect_profiles.sh ${TAG}_plane_100m plane_100m
tools/collect_profiles.sh ${TAG}_plane_lidar plane_lidar
unset PCR_BENCH_NO_PMC
timeout 400 python tools/soak.py 150 > $o/${TAG}_soak.txt 2>&1; tail -3 $o/${TAG}_soak.txt
TAG=$TAG tools/collect_set_target_profiles.sh > $o/${TAG}_set_target.log 2>&1; tail -4 $o/${TAG}_set_target.log
for n in 1.06e6 1e7 1e8; do timeout 300 python tools/knn_time.py $n 15 5 2>&1 | tail -1; done > $o/${TAG}_knn_time.txt; cat $o/${TAG}_knn_time.txt
for what in index voxels scan normals; do
  rm -rf $o/prof_tl; root=$(pwd)
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format rocpd -d $root/$o/prof_tl -o r -- python $root/tools/build_timeline.py run $what 1.06e6 2>&1 | grep "host wall" > $root/$o/${TAG}_timeline_${what}_after.txt)
  db=$(find $o/prof_tl -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/build_timeline.py show "$db" >> $o/${TAG}_timeline_${what}_after.txt 2>&1; fi
  rm -rf $o/prof_tl; tail -1 $o/${TAG}_timeline_${what}_after.txt
done
# (the million-pass rare-event trace of rounds 3-4 is not repeated: root-caused, docs/EXPERIMENTS.md)
# parity margins against the reference-run fixtures (worst max|dH|/max|H| per class and scan), for profiles/<tag>_g8_parity.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "g8_hip or g10_hip" 2>&1 | grep -oE "g8 [a-z0-9]+: worst.*|g10 [a-z]+: worst.*|[0-9]+ passed.*|[0-9]+ failed.*" > $o/${TAG}_g8_parity.txt; cat $o/${TAG}_g8_parity.txt

