cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4e
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pt_cache.pt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
cd /tmp
PARSENET_MS_SPARSE=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o b -- python3 $R/bench.py --profile-only > $O/po_k.json 2> $O/po_k.err
PARSENET_MS_SPARSE=1 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -o p -- python3 $R/bench.py --profile-only > $O/po_sq.json 2> $O/po_sq.err
cd $R
find $O -name "*kernel_trace.csv" -delete
K=$(find $O/k -name "b_kernel_stats.csv" | head -1); C=$(find $O/sq -name "p_counter_collection.csv" | head -1)
python tools/roofline_check.py $O/po_k.json $K $C $O/roofline_check.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4e"
for f in ("bench_cfg5","po_k"):
    d=json.loads([l for l in open(O+"/%s.json"%f).read().splitlines() if l.startswith("{")][-1])
    r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], d.get("value_dense"), r["kernel"], round(r["frac"],4), {k:(round(v["frac"],4), round(v["avg_launch_ms"],4), round(v["share_of_dense_pairs"],4)) for k,v in r["passes"].items()})
PY
