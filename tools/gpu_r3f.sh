#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python tools/torchprof.py cfg5 > $O/torchprof.log 2>&1
grep -v "^void pn_\|Cijk" $O/torchprof.log | head -70 | cut -c1-230
