#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
for st in 10 40; do
timeout 900 python bench.py --workload cfg5 --steps $st --warmup 2 --no-cpu-baseline > $O/bench_cfg5_$st.json 2> $O/bench_cfg5_$st.err
done
tail -45 $O/pytest.log; cat $O/bench_cfg5_*.json; tail -5 $O/bench_cfg5_10.err
