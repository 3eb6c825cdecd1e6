#!/bin/bash
# K2 at 1e8 and 1.06 M points: two-at-a-time loop (novec) / four consecutive points per lane (vec_nont) / + non-temporal streams (shipped)
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linearize_street or masked_multivoxel or robustness or fuzz_against_oracle or quirk_q6 or reuse" 2>&1 | tail -3
for lib in shipped red_w2nt red_w3 red_w4 red_w4nt; do
  [ $lib = shipped ] && unset PCR_LIB || export PCR_LIB=$root/build/exp/libpcr_$lib.so
  for big in "" 100m; do
    for pose in 0 99; do
      echo "== lib=$lib $big pose=$pose"
      timeout 900 python tools/pose_passes_timed.py $pose $big 2>&1 | grep "^pose"
    done
  done
done 2>&1 | tee $out/r05h2_reduce_variants.txt
