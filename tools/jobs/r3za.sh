#!/bin/bash
# round 3, job za: the whole GPU suite + smoke on the final tree (the record for profiles/r03_gpu_suite.txt)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/evidence_r03
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -s --durations=10 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
grep -i "parity\|eval-mode\|passed\|failed\|^rc " $O/pytest.log | cut -c1-900; tail -2 $O/smoke.log
