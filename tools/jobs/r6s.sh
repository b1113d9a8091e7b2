#!/bin/bash
# round 6, call s: gather of the final sort with eight loads in flight: kernel timers (knn64 = cfg4's layer + the bandwidth), parity
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s; mkdir -p $O
timeout 600 python tools/kbench.py knn64 > $O/kbench_knn64.log 2>&1
timeout 900 python -m pytest tests/test_knn_gpu.py -q -x -m gpu > $O/pytest_knn.log 2>&1; echo "rc $?" >> $O/pytest_knn.log
cat $O/kbench_knn64.log | cut -c1-200; tail -3 $O/pytest_knn.log
