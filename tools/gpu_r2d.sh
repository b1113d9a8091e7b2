#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python tools/kbench.py fitting_batch > $O/kbench_fit.log 2>&1
timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
tail -15 $O/pytest.log; cat $O/kbench_fit.log; cat $O/bench_cfg5.json; tail -5 $O/bench_cfg5.err
