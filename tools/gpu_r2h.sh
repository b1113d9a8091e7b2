#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2h
mkdir -p $O
timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_chamfer_gpu.py -m gpu -q --durations=10 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 300 python tools/kbench.py chamfer > $O/kbench.log 2>&1
tail -60 $O/pytest.log; cat $O/kbench.log
