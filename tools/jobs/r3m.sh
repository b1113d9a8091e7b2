#!/bin/bash
# round 3, job m: Chamfer kernel with ping-pong scalar prefetch: parity + timing
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m
mkdir -p $O
timeout 600 python -m pytest tests/test_chamfer_gpu.py tests/test_fitting_batch_gpu.py tests/test_golden_gpu.py tests/test_workloads_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 300 python tools/kbench.py chamfer > $O/kbench.log 2>&1
timeout 200 python tools/fuzz.py 40 > $O/fuzz.log 2>&1
tail -2 $O/pytest.log; grep -v "amdgpu.ids\|Warn" $O/kbench.log | tail -6; tail -1 $O/fuzz.log
