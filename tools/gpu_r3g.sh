#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for i in 1 2 3; do
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 > $O/bench_cfg5_$i.json 2> $O/bench_cfg5_$i.err; cut -c1-200 $O/bench_cfg5_$i.json
done
