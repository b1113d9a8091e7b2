#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3u
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 600 python tools/torch_sites.py > gpurun_out/r3u/sites.log 2>&1; echo "rc $?"; grep -v "Warning\|warn\|amdgpu" gpurun_out/r3u/sites.log | tail -50
