#!/bin/bash
# round 3, job z: the two chaotic parity tests at their new default (800 pre-training steps), three runs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $O
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_parity_fullsize_bwd_gpu.py -m gpu -q -s -k "whole_e2e_step or training_loop" > $O/pytest_$i.log 2>&1
done
grep -h "parity:\|passed\|failed" $O/pytest_*.log | cut -c1-1400
