#!/bin/bash
# two-phase filter kernel (PCR_FILTER_DEFER=1): exactness, then per-pose search times of the voxel configs; what the empty points cost
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
PCR_FILTER_DEFER=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "centroid_filter or quirk_q6 or linearize_masked or linearize_street or fuzz_against_oracle" 2>&1 | tail -3
PCR_FILTER_DEFER=1 timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "10m_centroid or g10" 2>&1 | tail -3
timeout 900 python tools/empty_query_probe.py vplane_10m 2>&1 | grep -v "^/opt" | tee $out/r05t_empty_query.txt
for d in 0 1; do
  for cfg in vplane_10m ndt_10m; do
  echo "== PCR_FILTER_DEFER=$d $cfg"
  PCR_FILTER_DEFER=$d timeout 900 python tools/reuse_probe.py --config $cfg --reps 5 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total\|^align" | awk '{ if ($1=="pose") printf "%s(%s) ", $14, $16; else print }'
  done
done 2>&1 | tee $out/r05t_filter_defer.txt
