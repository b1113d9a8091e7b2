#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
timeout 900 python tools/dbg/knn_probe.py > gpurun_out/dbg/sparsity.log 2>&1
cat gpurun_out/dbg/sparsity.log
