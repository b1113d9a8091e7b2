#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python tools/host_cprofile.py > $O/cprof.log 2>&1
grep -v "^$" $O/cprof.log | head -130 | cut -c1-170
