#!/bin/bash
# round 6, session p: ONE scan cut into 2 / 4 concurrent shards on ONE GPU (single-process group with repeated device ids, strong scaling)
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
for cfg in plane_b01 icp_b01; do
timeout 300 python bench.py --config $cfg --no-pmc --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1 context', d['config']['workload'], d['value'], d['ms_per_step'], d['kernels'])" | tee -a $o/r06p_strong_1gpu.txt
for n in 2 3 4; do
  PCR_BENCH_GROUP_DEVICES=$(python -c "print(','.join(['0']*$n))") GPU_MAX_HW_QUEUES=$((n+4)) timeout 600 python bench.py --gpus $n --single-process --scaling strong --no-pmc --no-cpu-baseline --config $cfg --steps 20 --warmup 5 2> $o/r06p_$n.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n contexts, strong', d['config']['workload'], d['value'], d['ms_per_step'], d['kernels'], d['config'].get('scan_points_job'))" | tee -a $o/r06p_strong_1gpu.txt
done
done
