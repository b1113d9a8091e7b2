#!/bin/bash
# round-3 evidence in ONE gpurun call: full GPU suite, the bench lines (cfg5 default = the driver's
# command, cfg5 with forced planned launches, cfg4, cfg2, cfg3), rocprofv3 kernel statistics + step
# breakdown, PMC passes of the dense mean-shift kernels (FETCH_SIZE / WRITE_SIZE alone, SQ set),
# torch-side attribution and the host profile.  Condense afterwards: python tools/condense_r03.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence_r03
mkdir -p $O
(free -g | head -2; nproc; lscpu | grep "Model name") > $O/host.txt 2>&1
timeout 1700 python -m pytest tests -m gpu -q -s --durations=10 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
timeout 1200 python bench.py > $O/bench_cfg5.json 2> $O/bench_cfg5.err
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_b.json 2> $O/bench_cfg5_b.err
PARSENET_MS_SPARSE=1 timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_planned.json 2> $O/bench_cfg5_planned.err
timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python bench.py --workload cfg2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o b -- python3 $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-dense --profile-steps 0 > $O/prof5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s4 -o b -- python3 $R/bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > $O/prof4.log 2>&1
for W in cfg2 cfg3; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$W -o b -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > $O/prof_$W.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- python3 $R/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-dense --profile-steps 0 > $O/pmc_$C.log 2>&1
find $O/pmc_$C -name "*kernel_trace.csv" -delete
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_SQ -o p -- python3 $R/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-dense --profile-steps 0 > $O/pmc_SQ.log 2>&1
find $O/pmc_SQ -name "*kernel_trace.csv" -delete
cd $R
python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
python tools/condense_r02.py pmc r03_meanshift_x3_planned_cfg5_pmc.csv pn_ms3_kernel $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ; cp profiles/r03_meanshift_x3_planned_cfg5_pmc.csv $O/
timeout 600 python tools/torch_sites.py > $O/torch_sites.txt 2>&1
timeout 600 python tools/host_cprofile.py > $O/host_cprofile.txt 2>&1
tail -2 $O/smoke.log; cat $O/host.txt; grep -i "parity\|eval-mode\|passed\|failed\|^rc " $O/pytest.log | cut -c1-900; for f in bench_cfg5 bench_cfg5_b bench_cfg5_planned bench_cfg4 bench_cfg2 bench_cfg3; do cut -c1-330 $O/$f.json; done; cat $O/breakdown.txt; cat $O/r03_meanshift_x3_planned_cfg5_pmc.csv; grep -v "amdgpu.ids" $O/torch_sites.txt | head -16
