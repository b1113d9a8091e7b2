#!/bin/bash
# round 6, call t: the gather of the final sort with eight loads in flight against the tree before it (exported to _ab_prev/
# for this call only), alternating on one box: kernel timers of cfg4's 64-channel layer, bench lines of cfg4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6t; mkdir -p $O
for i in 1 2; do
  timeout 600 python tools/kbench.py knn64 > $O/kbench_new_$i.log 2>&1
  (cd _ab_prev && timeout 600 python tools/kbench.py knn64 > ../$O/kbench_prev_$i.log 2>&1)
done
for i in 1 2; do
  timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_new_$i.json 2> $O/bench_cfg4_new_$i.err
  (cd _ab_prev && timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline > ../$O/bench_cfg4_prev_$i.json 2> ../$O/bench_cfg4_prev_$i.err)
done
grep -H "knn_final\|^knn B\|^dot_sel" $O/kbench_*.log | cut -c1-160
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6t/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
