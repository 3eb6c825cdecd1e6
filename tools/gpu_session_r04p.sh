#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_reference_style.py -m gpu -x -q -k "vplane or ndt or voxel or centroid or filter or pinned or g8 or g2 or masked" 2>&1 | grep -E "passed|failed|Error" | tail -3
export PCR_BENCH_NO_RCCL_PROBE=1 PCR_BENCH_NO_PMC=1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/step', d['ms_per_step'], 'value', d['value'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'align', d['seam']['align_ms'], 'class align', d['seam'].get('class_align_from_host_array_ms'))
"; }
for c in vplane_10m ndt_10m vplane_b01_harness ndt_b01_harness; do
  for v in halo0.1 default halo0.1 default; do
    unset PCR_HALO; [ $v = halo0.1 ] && export PCR_HALO=0.1
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c $v"
  done
done 2>&1 | tee $o/r04p_filter_halo_bench.txt
