#!/bin/bash
# SQ counters of the k-NN kernels for the libraries given (paths relative to the repo; "default" = the shipped build)
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp; cd /tmp
for lib in "$@"; do
  rm -rf $o/prof_k
  if [ "$lib" != "default" ]; then export PCR_LIB=$root/$lib; else unset PCR_LIB; fi
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format rocpd -d $o/prof_k -o r -- python $root/tools/knn_time.py 1.06e6 15 > $o/prof_k.log 2>&1
  db=$(find $o/prof_k -name "*.db" | head -1)
  echo "== $lib"; if [ -n "$db" ]; then python $root/tools/rocpd_summary.py "$db" 2>&1 | grep "k_knn" | cut -c1-40,108-200 | grep -E "WAVE_CYCLES|INSTS_VALU|VMEM|SQ_WAVES"; fi
  rm -rf $o/prof_k
done > $o/r05z_knn_sq.txt 2>&1
cat $o/r05z_knn_sq.txt
