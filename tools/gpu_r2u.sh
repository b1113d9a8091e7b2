#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2u
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 0 > $O/prof.log 2>&1
echo "rocprof rc $?"
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
find $O/prof -name "*kernel_trace.csv" -delete
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    print("%8.3f ms  %6d calls  %8.1f us  %5.2f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:90]))
P
tail -2 $O/prof.log | cut -c1-200
