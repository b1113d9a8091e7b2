# plan chain: algebraic cosines in the pair sweeps, 18 halvings, caps out of the combining launch; wmax backward with lane masks
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5p
mkdir -p $O
timeout 1200 python -m pytest tests/test_fused_gpu.py tests/test_meanshift_gpu.py tests/test_determinism_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline > $O/b$i.json 2> $O/b$i.err
python - <<PY
import json
d=json.load(open("$O/b$i.json"))
k=d["kernels"] if "kernels" in d else d.get("kernel_ms_per_step", {})
print(round(d["value"],1), round(d["ms_per_step"],2), d["config"].get("tile_pairs_executed"), {n:k[n] for n in ("meanshift_fwd","weighted_max_bwd") if n in k})
PY
done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-dense --profile-steps 0 --steps 8 --warmup 3 > $O/bench_traced.json 2> $O/bench_traced.err
cd $GRAFT_REPO_ROOT
T=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
python tools/step_breakdown.py $T 5 detail > $O/breakdown_detail.txt 2>&1
rm -rf $O/tr
grep -A16 "mean-shift iterations" $O/breakdown_detail.txt | cut -c1-150
