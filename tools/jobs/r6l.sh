#!/bin/bash
# round 6, call l: evaluation-mode refit with up to 15 matching workers (8 before)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python tools/kbench.py eval > $O/kbench_eval.log 2>&1
timeout 600 python -m pytest tests/test_fitting_eval_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
grep "eval-mode" $O/kbench_eval.log | cut -c1-220; tail -3 $O/pytest.log
