#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_kernel_filter or large_step or align_matches or rccl_single" 2>&1 | tail -4
echo "== soak"; timeout 300 python tools/soak.py 150 2>&1 | grep -v "^/opt" | tail -6 | tee gpurun_out/r04_soak.txt
