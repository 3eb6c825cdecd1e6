#!/bin/bash
# peer-to-peer transport: what the exchange costs (1 rank attached; 2 ranks on the one GPU, weak)
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
timeout 600 python bench.py --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 GPU', d['value'], d['ms_per_step'], 'rccl_1rank', d.get('rccl_1rank'), 'p2p_1rank', d.get('p2p_1rank'))" | tee $out/r05v_p2p.txt
for t in p2p host; do
  echo "== 2 ranks on one GPU, PCR_COMM=$t"
  PCR_COMM=$t timeout 600 python bench.py --gpus 2 --backend gloo --config plane_b01 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['allreduce_transport'], d.get('per_rank_kernel_ms'))"
done 2>&1 | tee -a $out/r05v_p2p.txt
