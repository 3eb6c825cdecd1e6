#!/bin/bash
# round 6, session h (re-entry): whole GPU suite on the restored tree, default bench line, per-pose times of plane_b01 / plane_lidar, lidar work counters
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
S=$root/tools/gpu_session.sh
$S r06h tests
$S r06h bench:default
REPS=3 timeout 600 $S r06h poses:plane_lidar
REPS=6 timeout 600 $S r06h poses:plane_b01
(cd $root && PCR_LIB=$root/point_cloud_registration_amd/libpcr_hip_dev.so timeout 600 python tools/lb_counters_probe.py plane_lidar 2>&1 | grep -v "^/opt" | tee $o/r06_lidar_counters.txt)
BENCH_ARGS="--no-pmc --no-cpu-baseline --repeats 3" timeout 400 $S r06h bench:plane_lidar
BENCH_ARGS="--no-pmc --no-cpu-baseline --repeats 3" timeout 400 $S r06h bench:icp_lidar_harness
