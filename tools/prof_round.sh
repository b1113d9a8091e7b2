# Evidence of a round in one gpurun call: counters first (their summary feeds bench.py's
# roofline.traffic), then the bench lines, then the rocprofv3 kernel statistics.
# Afterwards, in the build container: python tools/condense_profiles.py
set -x
R=/root/repo/gpurun_out/r1b
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $R/pmc1 -o p -- python3 /root/repo/tools/kbench.py meanshift > $R/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/pmc2 -o p -- python3 /root/repo/tools/kbench.py meanshift > $R/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/pmc3 -o p -- python3 /root/repo/tools/kbench.py meanshift > $R/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/pmc4 -o p -- python3 /root/repo/tools/kbench.py meanshift > $R/pmc4.log 2>&1
cd /root/repo
python tools/condense_profiles.py --pmc-only
python bench.py --workload cfg5 --steps 10 --warmup 2 > $R/bench_cfg5.log 2>&1; tail -1 $R/bench_cfg5.log > $R/bench_cfg5.json
python bench.py --workload cfg4 --steps 20 --warmup 3 > $R/bench_cfg4.log 2>&1; tail -1 $R/bench_cfg4.log > $R/bench_cfg4.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/s5 -o b -- python3 /root/repo/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $R/s5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/s4 -o b -- python3 /root/repo/bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $R/s4.log 2>&1
ls $R
