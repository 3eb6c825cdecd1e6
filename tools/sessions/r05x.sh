#!/bin/bash
# index build with cell_start from the sorted ids: full parity suite, index timeline, seam probes
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $o/r05x_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/r05x_pytest_gpu.log; grep -E "passed|failed|rc=" $o/r05x_pytest_gpu.log | tail -3
cd /tmp
for what in index; do
  rm -rf $o/prof_tl
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format rocpd -d $o/prof_tl -o r -- python $root/tools/build_timeline.py run $what 1.06e6 2>&1 | grep "host wall" > $o/r05x_timeline_$what.txt
  db=$(find $o/prof_tl -name "*.db" | head -1)
  if [ -n "$db" ]; then python $root/tools/build_timeline.py show "$db" >> $o/r05x_timeline_$what.txt 2>&1; fi
  rm -rf $o/prof_tl
  cut -c1-110 $o/r05x_timeline_$what.txt
done
cd $root
timeout 300 python tools/set_target_probe.py 2>&1 | grep -v "^/opt" | head -6 > $o/r05x_seam_align.txt; timeout 200 python tools/align_seam_probe.py 2>&1 | grep -v '^/opt' | head -7 >> $o/r05x_seam_align.txt; cat $o/r05x_seam_align.txt
timeout 600 python tools/build_time.py 1.06e6 1e7 1e8 > $o/r05x_build_time.txt 2>&1; tail -3 $o/r05x_build_time.txt
