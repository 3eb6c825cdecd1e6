#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out; export TMPDIR=/tmp
export PCR_BENCH_NO_RCCL_PROBE=1 PCR_BENCH_NO_PMC=1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/step', d['ms_per_step'], 'value', d['value'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'align', d['seam']['align_ms'])
"; }
for c in icp_b01_harness plane_b01_100k; do
  for h in 0.1 0.25 0.4 0.1 0.25 0.4; do
    PCR_HALO=$h timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c halo=$h"
  done
done 2>&1 | tee $o/r04r_small_scan_halo.txt
