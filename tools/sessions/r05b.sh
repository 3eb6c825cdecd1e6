#!/bin/bash
# potential of a near-exact seed: 40 passes at one pose, each seeded with the previous (identical) pass' match
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
for lib in shipped seed; do
  [ $lib = seed ] && export PCR_LIB=$root/build/exp/libpcr_seed.so
  for pose in 0 1 2 4; do
    echo "== lib=$lib pose=$pose"
    timeout 600 python tools/pose_passes_timed.py $pose
  done
done 2>&1 | grep -v "^/opt" | tee $out/r05b_seed_potential.txt
