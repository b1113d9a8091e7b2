#!/bin/bash
# Round-6 evidence in ONE gpurun call.  Condense afterwards: python tools/condense_r06.py
#  * full GPU suite (+ smoke), the determinism probe on all four workloads;
#  * bench lines: cfg5 twice (the driver's command: same pre-training loss / clusters / tile pairs in both),
#    cfg4, cfg2, cfg3;
#  * rocprofv3 of `bench.py --profile-only` (ONLY the launches the roofline is quoted on): kernel statistics,
#    SQ counters, FETCH_SIZE and WRITE_SIZE in passes of their own; tools/roofline_check.py;
#  * kernel statistics of cfg4 / cfg2 / cfg3, step breakdown of cfg5, the named kernels (kbench).
# EVIDENCE_SHORT=1: tests, determinism, bench lines and the default profile-only passes only (a late change that
# touches one kernel: the other passes of the previous full run stay valid).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence_r06
mkdir -p $O
(free -g | head -2; nproc; lscpu | grep "Model name") > $O/host.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q -s --durations=10 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
for W in cfg2 cfg3 cfg4; do timeout 300 python tools/determinism_probe.py --workload $W --steps 3 > $O/det_$W.txt 2>&1; done
timeout 600 python tools/determinism_probe.py --workload cfg5 --pretrain 40 --steps 3 > $O/det_cfg5.txt 2>&1
timeout 1200 python bench.py > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 1200 python bench.py --no-cpu-baseline > $O/bench_cfg5_b.json 2> $O/bench_cfg5_b.err
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r06.pt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cfg5_c.json 2> $O/bench_cfg5_c.err
timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python bench.py --workload cfg2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
cd /tmp
# (a) the default: mean-shift backward through the centre rows only — of the matrix-core mean-shift kernels only the
#     forward pass runs; kernel statistics + SQ counters of two --profile-only processes
PARSENET_MS_SPARSE=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o b -- python3 $R/bench.py --profile-only > $O/po_stats.json 2> $O/po_stats.err
cp $(find $O/s5 -name "b_kernel_trace.csv" | head -1) $O/s5_trace.csv 2>/dev/null
PARSENET_MS_SPARSE=1 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_SQ_rows -o p -- python3 $R/bench.py --profile-only > $O/po_sq_rows.json 2> $O/po_sq_rows.err
# (HBM traffic of exactly those launches — the forward-only plans at 1e-6 — in passes of their own: what `roofline.traffic`
#  of the default line quotes)
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
PARSENET_MS_SPARSE=1 timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${C}_rows -o p -- python3 $R/bench.py --profile-only > /dev/null 2> $O/po_${C}_rows.err
done
if [ -z "$EVIDENCE_SHORT" ]; then
# (the dense backward passes — PARSENET_MS_ROWS_BWD=0, callers with a dense gradient — are unchanged since round 5:
#  profiles/r05_*dense* stay valid)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s4 -o b -- python3 $R/bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > $O/prof4.log 2>&1
for W in cfg2 cfg3; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$W -o b -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > $O/prof_$W.log 2>&1
done
fi
cd $R
K=$(find $O/s5 -name "b_kernel_stats.csv" | head -1); C=$(find $O/pmc_SQ_rows -name "p_counter_collection.csv" | head -1)
python tools/roofline_check.py $O/po_stats.json $K $C $O/roofline_check.txt > /dev/null
python tools/step_breakdown.py $O/s5_trace.csv > $O/breakdown_profile_only.txt 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o b -- python3 $R/bench.py --no-cpu-baseline --no-dense --profile-steps 0 --steps 8 --warmup 3 > $O/bench_traced.json 2> $O/bench_traced.err
cd $R
T=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
python tools/step_timeline.py $T 1 1 > $O/timeline.txt 2>&1
python tools/step_breakdown.py $T 5 > $O/breakdown.txt 2>&1
python tools/step_gaps.py $T 2 > $O/gaps.txt 2>&1
rm -rf $O/tr
if [ -z "$EVIDENCE_SHORT" ]; then
cd /tmp
for CS in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
D=$O/pmc_named_$(echo $CS | cut -d' ' -f1)
timeout 600 rocprofv3 --pmc $CS --kernel-trace --output-format csv -d $D -o p -- python3 $R/tools/kbench.py edge smallk > /dev/null 2>$D.err
done
cd $R
fi
rm -f $O/s5_trace.csv
find $O -name "*kernel_trace.csv" -delete
if [ -z "$EVIDENCE_SHORT" ]; then
timeout 900 python tools/kbench.py edge chamfer smallk wgrad eval > $O/kbench.log 2>&1
timeout 300 python tools/probes/clock_under_load.py 4 > $O/clock_sysfs.txt 2>&1
# the data-parallel machinery forced on the one rank (RCCL process group, broadcast, all-reduce, barriers)
PARSENET_FORCE_COLLECTIVE=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_forced_collective.json 2> $O/bench_cfg5_forced_collective.err
timeout 600 python tools/torch_sites.py > $O/torch_sites.txt 2>&1
timeout 600 python tools/host_cprofile.py > $O/host_cprofile.txt 2>&1
timeout 600 python tools/probes/op_lines.py > $O/op_lines.txt 2>&1
timeout 600 python tools/probes/copy_census.py > $O/copy_census.txt 2>&1
fi
tail -2 $O/smoke.log; cat $O/host.txt; grep -i "passed\|failed\|^rc " $O/pytest.log | cut -c1-300; tail -n 1 $O/det_cfg*.txt
for f in bench_cfg5 bench_cfg5_b bench_cfg5_c bench_cfg4 bench_cfg2 bench_cfg3; do cut -c1-260 $O/$f.json; done
cat $O/roofline_check.txt; cat $O/breakdown.txt
