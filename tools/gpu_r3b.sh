#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; cut -c1-250 $O/bench_cfg5.json
