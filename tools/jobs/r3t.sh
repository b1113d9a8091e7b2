#!/bin/bash
# round 3, job t: the final ping-pong configuration: mean-shift tests (incl. bit identity across schedules) + bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3t
mkdir -p $O
timeout 1500 python -m pytest tests/test_meanshift_gpu.py tests/test_fullsize_gpu.py tests/test_parity_fullsize_bwd_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pre_cfg5.pt
for pp in 1 0 1 0; do
  PN_MS_PINGPONG=$pp timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > $O/bench_pp${pp}_$RANDOM.json 2> $O/bench_pp$pp.err
done
