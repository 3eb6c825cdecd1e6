#!/bin/bash
# round 3: full GPU suite, bench lines (default + the non-copy workloads)
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03b_pytest.log; tail -4 gpurun_out/r03b_pytest.log
timeout 600 python bench.py > gpurun_out/r03b_bench_plane_b01.json 2> gpurun_out/r03b_bench_plane_b01.err
tail -c 3000 gpurun_out/r03b_bench_plane_b01.json
for c in plane_b01_resampled plane_b01_crop; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > gpurun_out/r03b_bench_$c.json 2> gpurun_out/r03b_bench_$c.err
done
