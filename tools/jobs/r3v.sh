#!/bin/bash
# round 3, job v: forward pass with spread DMA (PN_MS_PINGPONG=1) vs forward pass on the round-2 schedule (3)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3v
mkdir -p $O
timeout 900 python -m pytest tests/test_meanshift_gpu.py -m gpu -q -x -k "block_sparse or iterations_forward or split_backward or pingpong" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pre_cfg5.pt
for pp in 1 3 1 3; do
  PN_MS_PINGPONG=$pp timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > $O/bench_pp${pp}_$RANDOM.json 2> $O/bench_pp$pp.err
done
PN_EXTRA_HIPCC_FLAGS=-DMS_TIMING python -m parsenet_codebase_amd.build > $O/build.log 2>&1
for pp in 1 3; do
PN_MS_PINGPONG=$pp PARSENET_MS_SPARSE=0 timeout 300 python tools/ms_timing.py > $O/timing_dense_pp$pp.txt 2>&1
grep PASS $O/timing_dense_pp$pp.txt
done
