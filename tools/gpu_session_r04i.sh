#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "knn or normals or voxel_build or deeper" 2>&1 | tail -3
timeout 300 python tools/set_target_probe.py 2>&1 | grep -v "^/opt" | head -7
timeout 300 python tools/set_target_profile.py 1.06e6 12 2>/dev/null | tail -1
timeout 300 python tools/set_target_profile.py 1e7 6 2>/dev/null | tail -1
