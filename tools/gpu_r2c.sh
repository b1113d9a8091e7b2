#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
timeout 900 python -m pytest tests/test_fitting_batch_gpu.py -m gpu -q > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_golden_gpu.py tests/test_workloads_gpu.py tests/test_trainer_gpu.py tests/test_chamfer_gpu.py tests/test_fitting_gpu.py -m gpu -q > $O/pytest_e2e.log 2>&1; echo "rc $?" >> $O/pytest_e2e.log
timeout 300 python tools/kbench.py fitting > $O/kbench_fit.log 2>&1
tail -40 $O/pytest_new.log; tail -40 $O/pytest_e2e.log; cat $O/kbench_fit.log
