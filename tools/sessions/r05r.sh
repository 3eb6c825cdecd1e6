#!/bin/bash
# Morton-ordered gather copy of the PlaneICP records (targets >= 4 M points): exactness, then k_reduce_finalize / k_nn_scan at 1e8 points with and without it
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "morton_gather or tile_handout or deeper_list or linearize_street" 2>&1 | tail -3
for v in "off:PCR_MORTON_GATHER_MIN=-1" "on:PCR_MORTON_GATHER_MIN=4000000"; do
  name=${v%%:*}; e=${v#*:}
  for pose in 0 12 99; do
    echo "== morton gather $name, plane_100m pose $pose"
    env $e timeout 900 python tools/pose_passes_timed.py $pose 100m 2>&1 | grep "^pose"
  done
done 2>&1 | tee $out/r05r_morton_gather.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "100m" 2>&1 | tail -3
