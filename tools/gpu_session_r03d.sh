#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "align or certified or fuzz_against or rccl or masked" > gpurun_out/r03d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03d_pytest.log; tail -3 gpurun_out/r03d_pytest.log
for c in plane_b01 plane_b01_resampled icp_b01; do
  timeout 600 python tools/reuse_probe.py --config $c --reps 10 --modes 1 --tol 1e-3 2>&1 | grep "pose\|total"
  timeout 300 python tools/pose_profile.py --config $c --modes 0 --reps 5 --brief 2>&1 | grep "align\|walk"
done
timeout 600 python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels'], d['seam'])"
