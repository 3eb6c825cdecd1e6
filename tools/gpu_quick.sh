#!/bin/bash
# quick A/B on the GPU: exactness subset + per-pose times of the default config(s)
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz_against_oracle or nn_stress or nn_query or certified_reuse or fuzz_knn" > gpurun_out/quick_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/quick_pytest.log; tail -3 gpurun_out/quick_pytest.log
for c in ${CONFIGS:-plane_b01}; do
  timeout 900 python tools/reuse_probe.py --config $c --reps ${REPS:-10} --modes 0 --tol 1e-3 2>&1 | grep -v "^/opt" > gpurun_out/quick_probe_$c.txt
  cat gpurun_out/quick_probe_$c.txt
done
