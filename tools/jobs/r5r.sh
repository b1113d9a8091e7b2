# small-k: eight waves for the 256-channel instance, union threshold: full kNN tests + cfg5 / cfg3 / cfg2 lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5r
mkdir -p $O
timeout 1200 python -m pytest tests/test_knn_gpu.py tests/test_encoders_gpu.py tests/test_determinism_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline > $O/b$i.json 2> $O/b$i.err
python - <<PY
import json
d=json.load(open("$O/b$i.json"))
print(round(d["value"],1), round(d["ms_per_step"],2), d["roofline"].get("frac"))
PY
done
for W in cfg3 cfg2 cfg4; do timeout 300 python bench.py --workload $W --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W', round(d['value'],1), round(d['ms_per_step'],3))"; done
