# A/B of the round-3 tree (.ab_old) against the working tree on ONE box, alternating
for rep in 1 2 3; do
  for wl in cfg2 cfg4; do
    (cd .ab_old && python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', '$wl', round(d['value'],1), round(d['ms_per_step'],3))")
    python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', '$wl', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
