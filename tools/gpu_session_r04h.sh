#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "deeper_list or voxel_build or centroid_filter or pinned" 2>&1 | tail -3
export PCR_BENCH_NO_RCCL_PROBE=1 PCR_BENCH_NO_PMC=1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/step', d['ms_per_step'], 'min', d['ms_per_step_min'], {k: v['avg_ms'] for k, v in d['kernels'].items()})
"; }
export PCR_LIB=$PWD/point_cloud_registration_amd/libpcr_hip_dev.so
for c in vplane_10m ndt_10m; do
  for dbg in 0 1 2; do
    PCR_FIX_DEBUG=$dbg timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c FIX_DEBUG=$dbg"
  done
done 2>&1 | tee $o/r04h_fix_debug.txt
unset PCR_LIB
for c in vplane_10m ndt_10m; do
  for v in base new base new; do
    unset PCR_LIB
    [ $v = base ] && export PCR_LIB=$PWD/build/exp/libpcr_base.so
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c $v"
  done
done 2>&1 | tee $o/r04h_ab_voxel.txt
