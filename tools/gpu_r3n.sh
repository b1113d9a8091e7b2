#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $O
PN_KNN_X3=2 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s4 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $O/prof4.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
python - <<'P'
import csv, glob
f = glob.glob("gpurun_out/r3n/s4/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print("%9.3f ms/step %6.1f calls/step %9.1f us avg %9.1f max  %s" % (float(r["TotalDurationNs"]) / 1e6 / 12, int(r["Calls"]) / 12, float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Name"][:70]))
P
python - <<'P'
import csv, glob
f = glob.glob("gpurun_out/r3n/s4/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
print("kNN ms/step:", sum(float(r["TotalDurationNs"]) for r in rows if "knn" in r["Name"]) / 1e6 / 12)
for r in rows:
    if "knn" in r["Name"]: print("%8.3f ms/step %5.1f calls %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / 12, int(r["Calls"]) / 12, float(r["AverageNs"]) / 1e3, r["Name"][:60]))
P
