#!/bin/bash
# round 6, session r: the whole GPU suite, smoke and the default + lidar bench lines on the final library
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q -rs --durations=15 > $o/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/r06_pytest_gpu.log; tail -4 $o/r06_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $o/r06_smoke.txt
for c in plane_b01 plane_lidar icp_lidar_harness; do
  timeout 900 python bench.py --config $c > $o/r06_bench_$c.json 2> $o/r06_bench_$c.err; cut -c1-260 $o/r06_bench_$c.json
done
