#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2w
mkdir -p $O
timeout 300 python tools/kbench.py knn64 > $O/kbench.log 2>&1; cat $O/kbench.log | tail -30
timeout 1200 python -m pytest tests/test_knn_gpu.py tests/test_fullsize_gpu.py tests/test_golden_gpu.py tests/test_meanshift_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
