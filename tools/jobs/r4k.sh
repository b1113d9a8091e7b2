cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4k
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pt_cache.pt
python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o b -- python3 $R/bench.py --profile-only > $O/po.json 2> $O/po.err
cd $R
find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, json, os, glob
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4k"
d=json.loads([l for l in open(O+"/bench.json").read().splitlines() if l.startswith("{")][-1])
print("bench", round(d["value"],2), round(d["ms_per_step"],2))
f=glob.glob(O+"/k/**/b_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
n=sum(int(r["Calls"]) for r in rows)
print("profile-only trace: %.1f ms, %d launches (pre-training from cache + 4 profiled steps)" % (tot/1e6, n))
for r in rows[:45]:
    print("%8.3f ms %6d  %7.1f us  %s" % (float(r["TotalDurationNs"])/1e6/4, int(r["Calls"])/4, float(r["TotalDurationNs"])/int(r["Calls"])/1e3, r["Name"][:100]))
PY
