#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2r
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 python bench.py --workload cfg5 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_cfg5_40.json 2> $O/bench_cfg5_40.err
tail -5 $O/pytest.log; cut -c1-200 $O/bench_cfg5.json; cut -c1-200 $O/bench_cfg5_40.json
