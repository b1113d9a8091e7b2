#!/bin/bash
# round 3, job zf: last check of the final tree — the driver's three commands (GPU suite with -x, smoke, bench.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3zf
mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; echo "rc $?" >> $O/bench.err
tail -3 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-600 $O/bench.json; cat $O/bench_time.txt
