#!/bin/bash
# round 3, job w: the two chaotic full-size parity tests, twice on the round-2 schedule and twice on the default one
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3w
mkdir -p $O
for pp in 0 1 0 1; do
PN_MS_PINGPONG=$pp timeout 900 python -m pytest tests/test_parity_fullsize_bwd_gpu.py -m gpu -q -s -k "whole_e2e_step or training_loop" > $O/pytest_pp${pp}_$RANDOM.log 2>&1
done
grep -h "parity:\|passed\|failed" $O/pytest_pp*.log | cut -c1-1200
