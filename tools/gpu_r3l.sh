#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 300 python tools/kbench.py knn64 > $O/kbench.log 2>&1; grep -v amdgpu $O/kbench.log | head -12
timeout 1500 python -m pytest tests/test_knn_gpu.py tests/test_fullsize_gpu.py tests/test_golden_gpu.py tests/test_encoder_gpu.py tests/test_edgeconv_gpu.py tests/test_meanshift_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -6 $O/pytest.log
