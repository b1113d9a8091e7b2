#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
timeout 900 python -m pytest tests/test_fitting_batch_gpu.py tests/test_chamfer_gpu.py tests/test_fitting_gpu.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
timeout 300 python tools/kbench.py chamfer > $O/kbench.log 2>&1
for q in 1 2 4; do PN_CHAMFER_Q=$q timeout 300 python tools/kbench.py chamfer >> $O/kbench.log 2>&1; done
timeout 600 python tools/pretrain_probe.py > $O/probe.log 2>&1
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_golden_gpu.py tests/test_workloads_gpu.py tests/test_trainer_gpu.py -m gpu -x -q > $O/pytest_e2e.log 2>&1; echo "rc $?" >> $O/pytest_e2e.log
timeout 300 python tools/kbench.py fitting > $O/kbench_fit.log 2>&1
tail -25 $O/pytest_new.log; cat $O/kbench.log; cat $O/probe.log; tail -25 $O/pytest_e2e.log; cat $O/kbench_fit.log
