#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2f
mkdir -p $O
timeout 600 python -m pytest tests/test_fitting_batch_gpu.py tests/test_golden_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/s5 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --pretrain 300 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1
find $O/s5 -name "*kernel_trace.csv" -delete
tail -8 $O/pytest.log; cat $O/breakdown.txt; tail -2 $O/prof.log | cut -c1-600
