#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3m
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 300 python tools/kbench.py knn64 > $O/kbench.log 2>&1; grep -v amdgpu $O/kbench.log | head -18
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-200 $O/bench_cfg4.json
for i in 1 2; do
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg5_$i.json 2> $O/bench_cfg5_$i.err; cut -c1-200 $O/bench_cfg5_$i.json
done
