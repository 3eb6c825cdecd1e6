#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel-trace + FETCH_SIZE / WRITE_SIZE passes of the set_target-side builds at 1.06 M / 10 M / 1e8 points.
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out; export TMPDIR=/tmp; cd /tmp
for spec in "1.06e6 12" "1e7 6" "1e8 3"; do
  set -- $spec; n=$1; reps=$2
  tag=${TAG:-r05}_set_target_$n
  python $root/tools/set_target_profile.py $n $reps 2>/dev/null | tail -1 > $out/${tag}_wall.txt; cat $out/${tag}_wall.txt
  for pass in stats fetch write; do
    rm -rf $out/prof_st
    case $pass in stats) args="--stats";; fetch) args="--pmc FETCH_SIZE";; write) args="--pmc WRITE_SIZE";; esac
    timeout 900 rocprofv3 $args --kernel-trace --output-format rocpd -d $out/prof_st -o r -- python $root/tools/set_target_profile.py $n $reps > $out/prof_st.log 2>&1
    db=$(find $out/prof_st -name "*.db" | head -1)
    python $root/tools/rocpd_summary.py "$db" > $out/${tag}_$pass.txt 2>&1
    rm -rf $out/prof_st
  done
  head -25 $out/${tag}_stats.txt
done
