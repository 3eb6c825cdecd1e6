#!/bin/bash
# first run of the MFMA-filtered search: exactness (every pipeline test incl. "mfma"), then per-pose times forced on (nn_mode 4) vs off
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfma or fuzz_against_oracle or nn_stress" > $out/r05e_pytest.log 2>&1; echo "rc=$?" >> $out/r05e_pytest.log; tail -15 $out/r05e_pytest.log
for mode in 5 4; do
  for cfg in plane_b01 plane_b01_resampled; do
  echo "== PCR_NN_MODE=$mode $cfg: nn us per pose"
  PCR_NN_MODE=$mode timeout 600 python tools/reuse_probe.py --config $cfg --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total\|identical" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
  done
done 2>&1 | tee $out/r05e_mfma_per_pose.txt
