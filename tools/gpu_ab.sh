#!/bin/bash
# A/B of library variants on the GPU: LIBS="name=path ..." CONFIGS="..." REPS=n
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
for spec in $LIBS; do
  name=${spec%%=*}; path=${spec#*=}
  for c in ${CONFIGS:-plane_b01}; do
    echo "=== $name $c"
    PCR_LIB=$path timeout 900 python tools/reuse_probe.py --config $c --reps ${REPS:-10} --modes 0 --tol 1e-3 2>&1 | grep -v "^/opt" | grep "pose\|total\|align" | tee gpurun_out/ab_${name}_$c.txt
  done
done
