#!/bin/bash
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = default ]; then unset PCR_LIB; else export PCR_LIB=$root/$lib; fi
  for n in 1.06e6 1e7 1e8; do timeout 300 python tools/knn_time.py $n 15 5 2>&1 | tail -1; done
done > $o/r05y2_knn_variants.txt; cat $o/r05y2_knn_variants.txt
