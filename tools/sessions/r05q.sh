#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz_against_oracle or nn_stress or linearize_street or align_matches or rccl_single" 2>&1 | tail -3
for cfg in plane_b01 icp_b01 plane_b01_resampled; do
  echo "== $cfg"
  timeout 600 python tools/reuse_probe.py --config $cfg --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total\|^align" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
done 2>&1 | tee $out/r05q_after_policy.txt
timeout 600 python bench.py --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench plane_b01', d['value'], d['ms_per_step'], d['kernels'], d['config'].get('first_align_ms'), d['config'].get('set_target_ms'))" | tee -a $out/r05q_after_policy.txt
for n in 3e6 5e6; do echo "scan size sweep TODO"; done > /dev/null
