#!/bin/bash
# round 6, call q: one-pass bf16 x 3 graph for small k (PN_KNN_FUSED=1): parity, kernel timings, A/B bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6q; mkdir -p $O
timeout 900 python -m pytest tests/test_knn_gpu.py -q -x -m gpu -k "one_pass_bf16 or small_k or near_ties or odd_shapes" > $O/pytest_knn.log 2>&1; echo "rc $?" >> $O/pytest_knn.log
tail -5 $O/pytest_knn.log
timeout 600 python tools/kbench.py smallk > $O/kbench_fused.log 2>&1
grep -n "knn B=" -B2 $O/kbench_fused.log | cut -c1-260 | head -60
