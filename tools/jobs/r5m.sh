cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5m
mkdir -p $O
python -m pytest tests/test_chamfer_gpu.py tests/test_golden_gpu.py tests/test_fitting_batch_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
python tools/kbench.py chamfer 2>&1 | grep chamfer
for Q in 1 2 4; do echo "PN_CHAMFER_Q=$Q"; PN_CHAMFER_Q=$Q python tools/kbench.py chamfer 2>&1 | grep "10000x10000\|1600x700"; done
