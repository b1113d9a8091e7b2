#!/bin/bash
# round 3, job x: the mean-shift kernels of commit 5f13491 against HEAD on one real embedding (same cached weights)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3x
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/w150.pt
timeout 600 python tools/cmp_x3_commits.py $O/new_a.npz > $O/new_a.log 2>&1
timeout 600 python tools/cmp_x3_commits.py $O/new_b.npz > $O/new_b.log 2>&1
PN_MS_PINGPONG=0 timeout 600 python tools/cmp_x3_commits.py $O/new_pp0.npz > $O/new_pp0.log 2>&1
cp parsenet_codebase_amd/csrc/meanshift_x3.h /tmp/x3_new.h
# (before the call: git show 5f13491:parsenet_codebase_amd/csrc/meanshift_x3.h > tools/dbg/meanshift_x3_r3o.h.txt)
cp tools/dbg/meanshift_x3_r3o.h.txt parsenet_codebase_amd/csrc/meanshift_x3.h
python -m parsenet_codebase_amd.build > $O/build_old.log 2>&1
timeout 600 python tools/cmp_x3_commits.py $O/old_a.npz > $O/old_a.log 2>&1
timeout 600 python tools/cmp_x3_commits.py $O/old_b.npz > $O/old_b.log 2>&1
tail -2 $O/*.log
python - <<'P'
import numpy as np
O="gpurun_out/r3x/"
f={k:dict(np.load(O+k+".npz")) for k in ("new_a","new_b","new_pp0","old_a","old_b")}
def cmp(a,b):
    print("==",a,"vs",b)
    for k in f[a]:
        x,y=f[a][k],f[b][k]
        if x.shape!=y.shape: print("  ",k,"shape",x.shape,y.shape); continue
        d=np.abs(x.astype(np.float64)-y.astype(np.float64)).max(); m=np.abs(y).max()
        print("   %-7s max|diff| %.3e  (max|ref| %.3e) equal %s"%(k,d,m,np.array_equal(x,y)))
cmp("new_a","new_b"); cmp("old_a","old_b"); cmp("new_a","old_a"); cmp("new_pp0","old_a")
P
