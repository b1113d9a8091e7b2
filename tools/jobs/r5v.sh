# bf16 x 3 GEMM on frozen weights only (new default) against rocBLAS everywhere: tests that involve the SplineNets, then
# alternating cfg5 lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5v
mkdir -p $O
timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_parity_fullsize_bwd_gpu.py tests/test_fitting_batch_gpu.py tests/test_fitting_eval_gpu.py tests/test_e2e_gpu.py tests/test_golden_gpu.py tests/test_fused_gpu.py tests/test_determinism_gpu.py -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r5v.pt
for i in 1 2 3; do for V in 0 frozen; do
PARSENET_GEMM_X3=$V timeout 600 python bench.py --no-cpu-baseline > $O/b_${V}_$i.json 2> $O/b_${V}_$i.err
python - <<PY
import json
d=json.load(open("$O/b_${V}_$i.json"))
print("gemm_x3=$V", round(d["value"],1), round(d["ms_per_step"],2), d["config"].get("clusters_per_shape"))
PY
done; done
