#!/bin/bash
# after making the chunk interleave compile-time (no scratch in any hot kernel): tests + the main bench lines
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or stress or tile_handout or centroid_filter or quirk_q6 or linearize or align_matches" 2>&1 | tail -3
for c in plane_b01 icp_b01 vplane_10m ndt_10m plane_100m; do
  timeout 900 python bench.py --config $c --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['config'].get('first_align_ms'), d['config'].get('set_target_ms'))"
done 2>&1 | tee $out/r05u_bench_check.txt
