#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2o
mkdir -p $O
timeout 1800 python -m pytest tests/test_fitting_batch_gpu.py tests/test_fullsize_gpu.py tests/test_meanshift_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
