#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfma or fuzz_against_oracle or nn_stress" > $out/r05f_pytest.log 2>&1; echo "rc=$?" >> $out/r05f_pytest.log; tail -5 $out/r05f_pytest.log
PCR_LIB=$root/build/exp/libpcr_mfstats.so timeout 600 python tools/mf_stats_probe.py plane_b01 2>&1 | grep -v "^/opt" | tee $out/r05f_mf_stats.txt
for mode in 4; do
  for cfg in plane_b01; do
  echo "== PCR_NN_MODE=$mode $cfg: nn us per pose"
  PCR_NN_MODE=$mode timeout 600 python tools/reuse_probe.py --config $cfg --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total\|identical" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
  done
done 2>&1 | tee $out/r05f_mfma_per_pose.txt
