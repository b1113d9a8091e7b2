#!/bin/bash
# round 3, job y: (1) old vs new mean-shift kernels incl. the one-shape 8 000-point path; (2) the two chaotic parity
# tests with PARITY_PRETRAIN steps of pre-training on their own shapes (cleaner modes), three runs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3y
mkdir -p $O
for i in 1 2 3; do
PARITY_PRETRAIN=600 timeout 900 python -m pytest tests/test_parity_fullsize_bwd_gpu.py -m gpu -q -s -k "whole_e2e_step or training_loop" > $O/pytest_600_$i.log 2>&1
done
grep -h "parity:\|passed\|failed" $O/pytest_600_*.log | cut -c1-1300
export PARSENET_PRETRAIN_CACHE=/tmp/w150.pt
timeout 600 python tools/cmp_x3_commits.py $O/new_a.npz > $O/new_a.log 2>&1
# (before the call: git show 5f13491:parsenet_codebase_amd/csrc/meanshift_x3.h > tools/dbg/meanshift_x3_r3o.h.txt)
cp tools/dbg/meanshift_x3_r3o.h.txt parsenet_codebase_amd/csrc/meanshift_x3.h
python -m parsenet_codebase_amd.build > $O/build_old.log 2>&1
timeout 600 python tools/cmp_x3_commits.py $O/old_a.npz > $O/old_a.log 2>&1
python - <<'P'
import numpy as np
O="gpurun_out/r3y/"
f={k:dict(np.load(O+k+".npz")) for k in ("new_a","old_a")}
for k in f["new_a"]:
    x,y=f["new_a"][k],f["old_a"][k]
    print("   %-7s max|diff| %.3e  (max|ref| %.3e) equal %s"%(k,np.abs(x.astype(np.float64)-y.astype(np.float64)).max(),np.abs(y).max(),np.array_equal(x,y)))
P
rm -f $O/*.npz
