# quick wins on latency-bound small kernels: tests + named kernel times inside a cfg5 step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5o
mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_meanshift_gpu.py tests/test_edgeconv_gpu.py tests/test_determinism_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline > $O/b$i.json 2> $O/b$i.err
python - <<PY
import json
d=json.load(open("$O/b$i.json"))
k=d["kernels"] if "kernels" in d else d.get("kernel_ms_per_step", {})
print(round(d["value"],1), round(d["ms_per_step"],2), {n:k[n] for n in ("meanshift_rows_bwd","weighted_max_bwd","weighted_max_fwd","edgeconv_reduce_fwd") if n in k})
PY
done
