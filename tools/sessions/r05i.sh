#!/bin/bash
# register top-k of the k-NN / normals kernels (PCR_KNN_REG=1, default) against the LDS list (=0); index build without the redundant synchronisations
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "knn or normals or nn_query or voxel_build or fuzz_against_oracle or stress" 2>&1 | tail -3
for reg in 0 1; do
  echo "== PCR_KNN_REG=$reg"
  PCR_KNN_REG=$reg timeout 600 python tools/build_time.py 1.06e6 1e7 2>&1 | grep -v "^/opt" | tail -8
done 2>&1 | tee $out/r05i_set_target.txt
timeout 300 python tools/set_target_probe.py 2>&1 | grep -v "^/opt" | head -8 | tee -a $out/r05i_set_target.txt
