#!/bin/bash
# round 3, job n: the whole GPU suite + smoke + the driver's bench command on the final tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc $?" >> $O/bench.err
tail -3 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-400 $O/bench.json; tail -2 $O/bench.err
