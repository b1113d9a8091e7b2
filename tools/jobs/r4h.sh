cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
export PARSENET_PRETRAIN_CACHE=/tmp/pt_cache.pt
for L in 2 1 3 4 6 2; do
PARSENET_MS_LLOYD=$L python bench.py --no-cpu-baseline --no-dense > gpurun_out/r4h/bench_$L.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4h/bench_$L.json").read().splitlines() if l.startswith("{")][-1])
r=d["roofline"]
print("lloyd $L", round(d["value"],2), round(d["ms_per_step"],2), {k:(round(v["avg_launch_ms"],3), round(v["share_of_dense_pairs"],3)) for k,v in r["passes"].items()})
PY
done
