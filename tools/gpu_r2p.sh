#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2p
mkdir -p $O
timeout 600 python tools/kbench.py fitting_batch > $O/kbench_fit.log 2>&1
timeout 300 python -m pytest tests/test_meanshift_gpu.py -m gpu -q > $O/pytest.log 2>&1
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/b.json 2>$O/b.err
grep -v "amdgpu.ids\|Warn\|warn" $O/kbench_fit.log; tail -2 $O/pytest.log; cut -c1-330 $O/b.json
