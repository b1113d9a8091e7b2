#!/bin/bash
# round-2 final evidence: counters first (their summary feeds roofline.traffic), full GPU suite,
# bench lines, rocprofv3 kernel statistics, step breakdown, named kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence_r02
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
# (the first run writes the pre-training cache)
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 > $O/first.json 2> $O/first.err
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- python3 $R/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $O/pmc_$C.log 2>&1
find $O/pmc_$C -name "*kernel_trace.csv" -delete
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_SQ -o p -- python3 $R/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $O/pmc_SQ.log 2>&1
find $O/pmc_SQ -name "*kernel_trace.csv" -delete
cd $R
python tools/condense_r02.py pmc r02_meanshift_x3_sparse_cfg5_pmc.csv pn_ms3_kernel $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ
cp profiles/r02_meanshift_x3_sparse_cfg5_pmc.csv $O/
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 python bench.py --workload cfg5 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_cfg5_40.json 2> $O/bench_cfg5_40.err
PARSENET_MS_SPARSE=0 timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_dense.json 2> $O/bench_cfg5_dense.err
PARSENET_MS_ARITH=f32 timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_f32.json 2> $O/bench_cfg5_f32.err
PARSENET_MS_ARITH=fp16x2 timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_fp16x2.json 2> $O/bench_cfg5_fp16x2.err
timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 300 python tools/kbench.py knn64 edge chamfer > $O/kbench.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o b -- python3 $R/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $O/prof5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s4 -o b -- python3 $R/bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $O/prof4.log 2>&1
cd $R
python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
tail -12 $O/pytest.log; for f in bench_cfg5 bench_cfg5_40 bench_cfg5_dense bench_cfg5_f32 bench_cfg5_fp16x2 bench_cfg4; do cut -c1-200 $O/$f.json; done; cat $O/breakdown.txt; cat $O/r02_meanshift_x3_sparse_cfg5_pmc.csv | head -8; grep -v "amdgpu.ids\|Warn\|warn" $O/kbench.log | tail -30
