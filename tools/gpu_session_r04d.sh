#!/bin/bash
# Round 4, session d: GPU suite; settle on / off on the 10 M voxel configs; the fused kernel's filter on the harness voxel configs.
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $o/r04d_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/r04d_pytest_gpu.log; tail -4 $o/r04d_pytest_gpu.log
export PCR_BENCH_NO_RCCL_PROBE=1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/step', d['ms_per_step'], 'min', d['ms_per_step_min'], 'noev', d['ms_per_step_events_off'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'align', d['seam']['align_ms'], 'class align', d['seam'].get('class_align_from_host_array_ms'), 'set_target', d['seam'].get('set_target_ms'))
"; }
for c in vplane_10m ndt_10m; do
  for v in base prologue settle base prologue settle; do
    unset PCR_LIB PCR_FILTER_SETTLE
    [ $v = base ] && export PCR_LIB=$PWD/build/exp/libpcr_base.so
    [ $v = settle ] && export PCR_FILTER_SETTLE=1
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c $v"
  done
done 2>&1 | tee $o/r04d_ab_settle.txt
for c in vplane_b01_harness ndt_b01_harness; do
  for v in f64 filter0 default f64 filter0; do
    unset PCR_LIB PCR_FILTER_AFTER PCR_VOX_FILTER
    [ $v = f64 ] && export PCR_VOX_FILTER=0
    [ $v = filter0 ] && export PCR_FILTER_AFTER=0
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c $v"
  done
done 2>&1 | tee $o/r04d_ab_fused_filter.txt
