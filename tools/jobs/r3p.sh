#!/bin/bash
# round 3, job p: in-kernel phase timers of the bf16 x 3 mean-shift passes (dense launches)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $O
PN_EXTRA_HIPCC_FLAGS=-DMS_TIMING python -m parsenet_codebase_amd.build > $O/build.log 2>&1
PARSENET_MS_SPARSE=0 timeout 300 python tools/ms_timing.py > $O/timing_dense.txt 2>&1
PARSENET_MS_SPARSE=1 timeout 300 python tools/ms_timing.py > $O/timing_planned.txt 2>&1
tail -3 $O/build.log; cat $O/timing_dense.txt $O/timing_planned.txt | grep -v amdgpu.ids
