#!/bin/bash
# round-2 evidence run after the block-sparse mean-shift: full GPU suite, bench lines, kernel stats, step breakdown, PMC
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2q
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 python bench.py --workload cfg5 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_cfg5_40.json 2> $O/bench_cfg5_40.err
PARSENET_MS_SPARSE=0 timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_dense.json 2> $O/bench_cfg5_dense.err
timeout 600 python bench.py --workload cfg4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python tools/kbench.py fitting_batch > $O/kbench_fit.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/s5 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1; find $O/s5 -name "*kernel_trace.csv" -delete; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$C -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_$C.log 2>&1
find $GRAFT_REPO_ROOT/$O/pmc_$C -name "*kernel_trace.csv" -delete
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_SQ -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_SQ.log 2>&1
find $GRAFT_REPO_ROOT/$O/pmc_SQ -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
tail -14 $O/pytest.log; for f in bench_cfg5 bench_cfg5_40 bench_cfg5_dense bench_cfg4; do cut -c1-330 $O/$f.json; done; cat $O/breakdown.txt; grep -v "amdgpu.ids\|Warn\|warn" $O/kbench_fit.log
