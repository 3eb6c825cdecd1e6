#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03c_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03c_pytest.log; tail -4 gpurun_out/r03c_pytest.log
for tl in 0 1; do
  echo "=== PCR_TILE_LOCAL=$tl"
  PCR_TILE_LOCAL=$tl timeout 600 python tools/reuse_probe.py --config plane_b01 --reps 10 --modes 0 --tol 1e-3 2>&1 | grep "pose\|total"
done
