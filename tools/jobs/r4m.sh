cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
export PARSENET_PRETRAIN_CACHE=/tmp/pt_cache.pt
for F in 384 0 384 0; do
PARSENET_MS_FINE=$F python bench.py --no-cpu-baseline --no-dense > gpurun_out/r4m/bench_$F.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4m/bench_$F.json").read().splitlines() if l.startswith("{")][-1])
r=d["roofline"]
print("fine $F", round(d["value"],2), round(d["ms_per_step"],2), {k:(round(v["avg_launch_ms"],3), round(v["share_of_dense_pairs"],3)) for k,v in r["passes"].items()}, d["config"]["clusters_per_shape"])
PY
done
timeout 900 python -m pytest tests/test_meanshift_gpu.py -x -q 2>&1 | tail -2
