#!/bin/bash
# neighbourhood row mask: exactness subset, then A/B (mask on / off by env on the same library; "plain" = the library before it)
cd "$(dirname "$0")/.."
o=gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "fuzz_against or stress or nn_query or b01_sampled or deeper or certified_reuse or g8 or 100m_plane" 2>&1 | grep -E "passed|failed|Error" | tail -3
for v in mask nomask plain; do
  unset PCR_LIB PCR_ROW_MASK
  [ $v = nomask ] && export PCR_ROW_MASK=0
  [ $v = plain ] && export PCR_LIB=$PWD/build/exp/libpcr_plain.so
  echo "== plane_b01 $v"
  timeout 600 python tools/reuse_probe.py --config plane_b01 --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
done 2>&1 | tee $o/r04m_rowmask_plane_b01.txt
export PCR_BENCH_NO_RCCL_PROBE=1 PCR_BENCH_NO_PMC=1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/step', d['ms_per_step'], 'min', d['ms_per_step_min'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'align', d['seam']['align_ms'], 'set_target', d['seam'].get('set_target_ms'))
"; }
for c in plane_b01 icp_b01 icp_b01_harness plane_b01_100k plane_b01_resampled; do
  for v in mask nomask; do
    unset PCR_ROW_MASK; [ $v = nomask ] && export PCR_ROW_MASK=0
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c $v"
  done
done 2>&1 | tee $o/r04m_rowmask_bench.txt
for v in mask nomask; do
  unset PCR_ROW_MASK; [ $v = nomask ] && export PCR_ROW_MASK=0
  echo "== plane_100m $v"
  timeout 900 python tools/reuse_probe.py --config plane_100m --reps 2 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total\|align" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
done 2>&1 | tee $o/r04m_rowmask_plane_100m.txt
unset PCR_ROW_MASK
timeout 300 python tools/set_target_probe.py 2>&1 | grep -v "^/opt" | head -7
