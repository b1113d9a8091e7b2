#!/bin/bash
# round-2 evidence run: bench lines + kernel stats (bf16x3 default, f32, fp16x2 opt-in) on the end-to-end steps only,
# cfg4 line + stats, mean-shift PMC traffic at the batched launch shape
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2i
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
for A in bf16x3 f32 fp16x2; do
  export PARSENET_MS_ARITH=$A
  timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_$A.json 2> $O/bench_cfg5_$A.err
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/s5_$A -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof_$A.log 2>&1)
  if [ $A = bf16x3 ]; then python tools/step_breakdown.py $O/s5_$A/b_kernel_trace.csv > $O/breakdown_$A.txt 2>&1; fi
  find $O/s5_$A -name "*kernel_trace.csv" -delete
done
unset PARSENET_MS_ARITH
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/s4 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof_cfg4.log 2>&1)
find $O/s4 -name "*kernel_trace.csv" -delete
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_ms1 -o p -- python3 $GRAFT_REPO_ROOT/tools/kbench.py meanshift_batch > $GRAFT_REPO_ROOT/$O/pmc_ms1.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_ms2 -o p -- python3 $GRAFT_REPO_ROOT/tools/kbench.py meanshift_batch > $GRAFT_REPO_ROOT/$O/pmc_ms2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_ms3 -o p -- python3 $GRAFT_REPO_ROOT/tools/kbench.py meanshift_batch > $GRAFT_REPO_ROOT/$O/pmc_ms3.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
cat $O/bench_cfg5.json $O/bench_cfg4.json $O/bench_cfg5_f32.json $O/bench_cfg5_fp16x2.json | cut -c1-1200; cat $O/breakdown_bf16x3.txt
