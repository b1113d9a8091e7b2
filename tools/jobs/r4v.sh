# per-point GEMMs on the bf16 matrix cores (PARSENET_GEMM_X3=1) against rocBLAS (0): tests, then the four workloads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4v
mkdir -p $O
python -m pytest tests/test_gemm_gpu.py tests/test_golden_gpu.py tests/test_encoder_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r04.pt
python bench.py --no-cpu-baseline --no-dense > /dev/null 2>&1
for V in 1 0 1 0 1 0; do
  for W in cfg5 cfg3 cfg4; do
    PARSENET_GEMM_X3=$V python bench.py --workload $W --no-cpu-baseline --no-dense > $O/b_${W}_$V.json 2> $O/b_${W}_$V.err
    echo "PARSENET_GEMM_X3=$V $W: $(python -c "import json;d=json.load(open('$O/b_${W}_$V.json'));print(round(d['value'],2), round(d['ms_per_step'],3))")"
  done
done > $O/ab.txt 2>&1
cat $O/ab.txt
