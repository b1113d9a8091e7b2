cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
export PARSENET_PRETRAIN_CACHE=/tmp/pt_cache.pt
for i in 1 2 3; do
python bench.py --no-cpu-baseline > gpurun_out/r4q/bench_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4q/bench_$i.json").read().splitlines() if l.startswith("{")][-1])
r=d["roofline"]
print("bench", round(d["value"],2), round(d["ms_per_step"],2), round(d.get("value_dense") or 0,2), {k:(round(v["frac"],3), round(v["avg_launch_ms"],3), round(v["share_of_dense_pairs"],3)) for k,v in r["passes"].items()}, d["config"]["clusters_per_shape"], {k:v for k,v in d["kernel_ms_per_step"].items() if "sel_" in k})
PY
done
unset PARSENET_PRETRAIN_CACHE
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 4
