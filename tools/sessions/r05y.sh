#!/bin/bash
# k-NN collect path: parity first, then timing of the variants given (default = shipped build)
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "knn or normal or g6 or g7 or g8_hip" > $o/r05y_pytest_knn.log 2>&1; echo "pytest rc=$?" >> $o/r05y_pytest_knn.log; tail -3 $o/r05y_pytest_knn.log
bash tools/sessions/r05y2.sh "$@"
bash tools/sessions/r05z.sh "$@"
