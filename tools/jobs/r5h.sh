cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5h
mkdir -p $O
python -m pytest tests/test_fitting_eval_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -40 $O/pytest.txt
python -m pytest tests/test_golden_gpu.py tests/test_fitting_gpu.py -x -q -m gpu > $O/pytest2.txt 2>&1
tail -5 $O/pytest2.txt
