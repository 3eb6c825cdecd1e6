#!/bin/bash
# round 6, session e: nearest-first box scans on the lidar configs, work counters (developer build), bench lines
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
S=$root/tools/gpu_session.sh
(cd $root && timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rs -k "lidar or g11" > $o/r06e_lidar.log 2>&1; echo "rc=$?" >> $o/r06e_lidar.log; tail -5 $o/r06e_lidar.log)
REPS=3 timeout 600 $S r06e poses:plane_lidar
PCR_GRID_CELL=0.2 REPS=3 timeout 600 $S r06e_cell0.2 poses:plane_lidar
PCR_HALO=0 REPS=3 timeout 600 $S r06e_halo0 poses:plane_lidar
(cd $root && PCR_LIB=$root/point_cloud_registration_amd/libpcr_hip_dev.so timeout 600 python tools/lb_counters_probe.py plane_lidar plane_b01 2>&1 | grep -v "^/opt" | tee $o/r06e_counters.txt)
BENCH_ARGS="--no-pmc --no-cpu-baseline --repeats 3" timeout 400 $S r06e bench:icp_lidar_harness
BENCH_ARGS="--no-cpu-baseline --repeats 3" timeout 600 $S r06e bench:plane_lidar
$S r06e quick
