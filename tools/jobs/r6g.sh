#!/bin/bash
# round 6, call g: the tree after the copy fixes: full GPU suite, smoke, two default bench processes back to back
# (is the first process of a fresh box slower?), cfg4 / cfg2 / cfg3 lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O
timeout 900 python bench.py > $O/bench_cfg5_first.json 2> $O/bench_cfg5_first.err
timeout 900 python bench.py > $O/bench_cfg5_second.json 2> $O/bench_cfg5_second.err
timeout 2400 python -m pytest tests -m gpu -q --durations=6 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_third.json 2> $O/bench_cfg5_third.err
timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --workload cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
tail -14 $O/pytest.log; tail -2 $O/smoke.log
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3))"; done
