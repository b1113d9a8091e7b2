cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5l
mkdir -p $O
python -m pytest tests/test_meanshift_gpu.py tests/test_determinism_gpu.py tests/test_e2e_gpu.py tests/test_fitting_batch_gpu.py tests/test_trainer_gpu.py tests/test_workloads_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
