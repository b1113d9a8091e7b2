cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5i
mkdir -p $O
python tools/kbench.py eval > $O/kbench_eval.txt 2>&1
tail -4 $O/kbench_eval.txt
python -m pytest tests/test_fitting_gpu.py tests/test_golden_gpu.py tests/test_fitting_eval_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
