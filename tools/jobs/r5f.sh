# round 5 checkpoint: the driver's bench command + the whole GPU suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5f
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r05.pt
python bench.py > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5f/bench_cfg5.json"))
print("cfg5", round(d["value"],2), "shapes/s", round(d["ms_per_step"],2), "ms; dense", d.get("value_dense"), "kernel ms:", {k: round(v,2) for k,v in sorted(d.get("kernel_ms_per_step",{}).items(), key=lambda kv:-kv[1])[:14]})
PY
python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
