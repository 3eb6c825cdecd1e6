#!/bin/bash
# Round 4, session b: whole GPU suite on the library with the pending-fix prologue in the reduce kernel; A/B of the voxel configs.
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $o/r04b_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/r04b_pytest_gpu.log; tail -4 $o/r04b_pytest_gpu.log
export PCR_BENCH_NO_RCCL_PROBE=1
for c in vplane_10m ndt_10m; do
  for lib in base new base new; do
    if [ $lib = base ]; then export PCR_LIB=$PWD/build/exp/libpcr_base.so; else unset PCR_LIB; fi
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c $lib', 'ms/step', d['ms_per_step'], 'min', d['ms_per_step_min'], 'noev', d['ms_per_step_events_off'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'align', d['seam']['align_ms'])
"
  done
done 2>&1 | tee $o/r04b_ab_fix.txt
