#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2n
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/b.json 2>$O/b.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/s5 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1; python tools/step_gaps.py $O/s5/b_kernel_trace.csv 2 > $O/gaps.txt 2>&1; find $O/s5 -name "*kernel_trace.csv" -delete; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$C -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_$C.log 2>&1
find $GRAFT_REPO_ROOT/$O/pmc_$C -name "*kernel_trace.csv" -delete
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_SQ -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_SQ.log 2>&1
find $GRAFT_REPO_ROOT/$O/pmc_SQ -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
tail -6 $O/pytest.log; cat $O/b.json | cut -c1-2500; cat $O/breakdown.txt; head -12 $O/gaps.txt
