#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $O
timeout 900 python -m pytest tests/test_meanshift_gpu.py -m gpu -x -q -k "caps or chain" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -25 $O/pytest.log
