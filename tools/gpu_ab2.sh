#!/bin/bash
# compact A/B: trajectory totals only.  LIBS="name=path ..." CONFIGS="..." REPS=n
cd /root/repo; export TMPDIR=/tmp
for c in ${CONFIGS:-plane_b01}; do
  for spec in $LIBS; do
    name=${spec%%=*}; path=${spec#*=}
    r=$(PCR_LIB=$path timeout 600 python tools/reuse_probe.py --config $c --reps ${REPS:-8} --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else printf "| nn total %s", $8 }')
    echo "$c $name: $r"
  done
done
