#!/bin/bash
# round 6, call d: the committed defaults (bf16 x 3 GEMM for frozen weights only, FlatAdam, one-launch gradient gather,
# cat-free conv5, Chamfer pre-filter with a 4-tile prefetch): Chamfer micro-benchmark, the FULL GPU suite, bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
timeout 600 python tools/kbench.py chamfer > $O/kbench.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --workload cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
cat $O/kbench.log | cut -c1-300; tail -25 $O/pytest.log
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3))"; done
