cd /root/repo
for c in plane_b01 icp_b01 plane_b01_resampled; do for h in 0.2 0.286 0.35 auto 0.47 0.57 0.8; do
  if [ $h = auto ]; then unset PCR_GRID_CELL; else export PCR_GRID_CELL=$h; fi
  timeout 200 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '$h', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done; done
