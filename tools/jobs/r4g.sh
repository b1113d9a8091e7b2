cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
export PARSENET_PRETRAIN_CACHE=/tmp/pt_cache.pt
python bench.py --no-cpu-baseline > gpurun_out/r4g/bench.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4g/bench.json").read().splitlines() if l.startswith("{")][-1])
print(round(d["value"],2), round(d["ms_per_step"],2))
tot=0
for k,(ms,n) in d["kernel_ms_per_step"].items():
    tot+=ms; print("%-24s %7.3f ms  %6.1f launches" % (k,ms,n))
print("sum of library kernels", round(tot,2))
PY
python tools/step_breakdown.py 2>/dev/null | tail -3
