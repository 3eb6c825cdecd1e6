#!/bin/bash
# round 6, session c: groups / N-rank p2p again; the heavy-cell index (leaf / group boxes) on the lidar configs, cell sweep
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
S=$root/tools/gpu_session.sh
(cd $root && timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_two_ranks.py -m gpu -q -rs > $o/r06c_group.log 2>&1; echo "rc=$?" >> $o/r06c_group.log; tail -15 $o/r06c_group.log)
(cd $root && timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rs -x -k "lidar" > $o/r06c_lidar.log 2>&1; echo "rc=$?" >> $o/r06c_lidar.log; tail -15 $o/r06c_lidar.log)
$S r06c quick
REPS=3 timeout 600 $S r06c poses:plane_lidar
for c in 0.2 0.3 0.4 0.6; do
  PCR_GRID_CELL=$c REPS=3 timeout 600 $S r06c_cell$c poses:plane_lidar
done
BENCH_ARGS="--no-pmc --no-cpu-baseline --repeats 3" timeout 400 $S r06c bench:icp_lidar_harness
REPS=5 $S r06c poses:plane_b01
