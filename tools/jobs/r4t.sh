cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
timeout 900 python -m pytest tests/test_meanshift_gpu.py tests/test_determinism_gpu.py -x -q 2>&1 | tail -n 2
export PARSENET_PRETRAIN_CACHE=/tmp/pt_cache.pt
for i in 1 2 3; do
python bench.py --no-cpu-baseline > gpurun_out/r4t/bench_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4t/bench_$i.json").read().splitlines() if l.startswith("{")][-1])
r=d["roofline"]
print("bench", round(d["value"],2), round(d["ms_per_step"],2), round(d.get("value_dense") or 0,2), {k:(round(v["frac"],3), round(v["avg_launch_ms"],3), round(v["share_of_dense_pairs"],3)) for k,v in r["passes"].items()}, d["config"]["clusters_per_shape"])
PY
done
