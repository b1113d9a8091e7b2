#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
timeout 900 python -m pytest tests/test_meanshift_gpu.py tests/test_fitting_batch_gpu.py tests/test_e2e_gpu.py tests/test_golden_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/b.json 2>$O/b.err
tail -5 $O/pytest.log; cut -c1-400 $O/b.json
