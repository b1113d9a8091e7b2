#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2y
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; cut -c1-250 $O/bench_cfg5.json
PN_KNN_X3=0 timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg5_nox3.json 2> $O/bench_cfg5_nox3.err; cut -c1-250 $O/bench_cfg5_nox3.json
