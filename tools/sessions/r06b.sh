#!/bin/bash
# round 6, session b: single-process groups + N = 2/4/8 peer-to-peer ranks; row-block boxes variant B (parallel loads) A/B; counters
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
S=$root/tools/gpu_session.sh
(cd $root && timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_two_ranks.py -m gpu -q -rs -x > $o/r06b_group.log 2>&1; echo "rc=$?" >> $o/r06b_group.log; tail -15 $o/r06b_group.log)
$S r06b quick
for rb in 0 1; do
  export PCR_RBOX=$rb
  REPS=10 $S r06b_rb$rb poses:plane_b01
  REPS=5 $S r06b_rb$rb poses:plane_b01_resampled
done
# counters of the first pose, boxes off / on (the null's evidence: L1 accesses and fabric bytes per query)
for rb in 0 1; do
  export PCR_RBOX=$rb
  $S r06b_rb$rb pmc:b01:0:TCP_TOTAL_CACHE_ACCESSES_sum,SQ_INSTS_VALU,SQ_INSTS_VMEM_RD
  $S r06b_rb$rb pmc:b01:0:FETCH_SIZE
done
