#!/bin/bash
# round 6, session a: row-block boxes A/B (PCR_RBOX=0|1 at target build) on the uniform configs; the non-uniform (lidar_sweep)
# configs on the round-5 index = the "before" of VERDICT r5 item 2
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
S=$root/tools/gpu_session.sh
$S r06a quick
for rb in 0 1; do
  export PCR_RBOX=$rb
  REPS=10 $S r06a_rb$rb poses:plane_b01
  REPS=5 $S r06a_rb$rb poses:plane_b01_resampled poses:plane_b01_crop poses:icp_b01
done
for rb in 0 1; do
  export PCR_RBOX=$rb
  REPS=3 timeout 600 $S r06a_rb$rb poses:plane_lidar
  BENCH_ARGS="--no-pmc --no-cpu-baseline --repeats 3" timeout 400 $S r06a_rb$rb bench:icp_lidar_harness
done
for rb in 0 1; do
  export PCR_RBOX=$rb
  REPS=2 timeout 900 $S r06a_rb$rb poses:plane_100m
done
