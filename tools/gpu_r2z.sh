#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2z
mkdir -p $O
timeout 900 python -m pytest tests/test_meanshift_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/pytest_ms.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest_ms.log
cd /tmp
MS_PROBE_EMB=$GRAFT_REPO_ROOT/tools/dbg/ms_emb.pt timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $GRAFT_REPO_ROOT/tools/ms_probe.py > $O/probe.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_trace.csv" -delete
head -12 $O/probe.log
python - <<'P'
import csv, glob
f = glob.glob("gpurun_out/r2z/prof/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "pn_ms" in r["Name"]:
        print("%9.1f us avg  %5d calls  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"][:70]))
P
