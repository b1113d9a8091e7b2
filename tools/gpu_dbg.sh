#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
timeout 600 python tools/dbg/cmp_fit.py 3000 > gpurun_out/dbg/cmp.log 2>&1
grep -A12 "bisect" gpurun_out/dbg/cmp.log; tail -3 gpurun_out/dbg/cmp.log
