#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3o
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 1800 python -m pytest tests/test_knn_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
for lvl in 1 2; do
PN_KNN_X3=$lvl timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_$lvl.json 2> $O/bench_cfg4.err; echo "cfg4 level $lvl"; cut -c60-200 $O/bench_cfg4_$lvl.json
PN_KNN_X3=$lvl timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg5_$lvl.json 2> $O/bench_cfg5.err; echo "cfg5 level $lvl"; cut -c60-200 $O/bench_cfg5_$lvl.json
done
