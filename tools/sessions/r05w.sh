#!/bin/bash
# device timelines of the set_target-side builds at 1.06 M (where does the wall-clock beyond the kernels go?)
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp; cd /tmp
for what in index normals voxels scan; do
  rm -rf $o/prof_tl
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format rocpd -d $o/prof_tl -o r -- python $root/tools/build_timeline.py run $what 1.06e6 2>&1 | grep "host wall" > $o/r05w_timeline_$what.txt
  db=$(find $o/prof_tl -name "*.db" | head -1)
  python $root/tools/build_timeline.py show "$db" >> $o/r05w_timeline_$what.txt 2>&1
  rm -rf $o/prof_tl
  echo "== $what"; head -3 $o/r05w_timeline_$what.txt; tail -1 $o/r05w_timeline_$what.txt
done
cd $root; timeout 200 python tools/build_timeline.py run index 1.06e6 2>&1 | grep "host wall"
