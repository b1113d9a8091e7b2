#!/bin/bash
# round 6, call a: the new parity / RCCL-on-one-GPU tests + the suites of the files whose kernels changed, one bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
timeout 1500 python -m pytest tests/test_rccl_world1_gpu.py tests/test_golden_gpu.py tests/test_fitting_eval_gpu.py tests/test_fused_gpu.py tests/test_knn_gpu.py -m gpu -q -s -x --durations=8 > $O/pytest_a.log 2>&1; echo "rc $?" >> $O/pytest_a.log
timeout 600 python -m pytest tests/test_meanshift_gpu.py -m gpu -q -x -k "other_embedding or centre_rows" > $O/pytest_b.log 2>&1; echo "rc $?" >> $O/pytest_b.log
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5.json 2> $O/bench_cfg5.err
tail -5 $O/pytest_a.log; tail -3 $O/pytest_b.log; cut -c1-400 $O/bench_cfg5.json
