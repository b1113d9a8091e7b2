#!/bin/bash
# round 6, call c: Chamfer pre-filter with shuffled operands, the weight gradient with the parallel bias reduction,
# FlatAdam, cat-free conv5: tests, micro-benchmarks, A/B bench lines; clock under load; re-pin search for the two
# whole-step tests that moved with the new rounding
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_chamfer_gpu.py tests/test_gemm_gpu.py tests/test_fused_gpu.py tests/test_encoder_gpu.py tests/test_workloads_gpu.py tests/test_golden_gpu.py tests/test_fitting_batch_gpu.py tests/test_determinism_gpu.py -m gpu -q > $O/pytest_a.log 2>&1; echo "rc $?" >> $O/pytest_a.log
timeout 600 python tools/kbench.py chamfer wgrad > $O/kbench.log 2>&1
timeout 300 python tools/probes/clock_under_load.py 4 > $O/clock_under_load.txt 2>&1
for rep in 1 2; do
PARSENET_PRETRAIN_CACHE=/tmp/pre_x3.pt timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_x3_$rep.json 2> $O/bench_cfg5_x3_$rep.err
PARSENET_PRETRAIN_CACHE=/tmp/pre_fr.pt PARSENET_GEMM_X3=frozen PARSENET_GEMM_X3_MIN_ROWS=512 PARSENET_FLAT_ADAM=0 PN_CHAMFER_MFMA=0 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_r5_$rep.json 2> $O/bench_cfg5_r5_$rep.err
done
timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_x3.json 2> $O/bench_cfg4_x3.err
PARSENET_GEMM_X3=frozen PARSENET_GEMM_X3_MIN_ROWS=512 PARSENET_FLAT_ADAM=0 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_r5.json 2> $O/bench_cfg4_r5.err
for W in cfg2 cfg3; do
timeout 600 python bench.py --workload $W --no-cpu-baseline > $O/bench_${W}_x3.json 2> $O/bench_${W}_x3.err
PARSENET_GEMM_X3=frozen PARSENET_GEMM_X3_MIN_ROWS=512 PARSENET_FLAT_ADAM=0 PN_CHAMFER_MFMA=0 timeout 600 python bench.py --workload $W --no-cpu-baseline > $O/bench_${W}_r5.json 2> $O/bench_${W}_r5.err
done
# re-pin search
for P in 600 700 800; do
PARITY_PRETRAIN=$P timeout 900 python -m pytest tests/test_parity_fullsize_bwd_gpu.py -q -s -m gpu -k "benchmark_size" > $O/whole_$P.txt 2>&1
grep "^.*whole-step parity" $O/whole_$P.txt | cut -c1-700; tail -1 $O/whole_$P.txt
done
for S in 56 13 51 55 58; do
PARITY_SPLINE_SHAPE=$S timeout 900 python -m pytest "tests/test_parity_fullsize_bwd_gpu.py::test_whole_e2e_step_with_a_cylinder_and_splines_on_pinned_graphs[56]" -q -s -m gpu > $O/pinned_$S.txt 2>&1
grep "pinned-graph whole step" $O/pinned_$S.txt | cut -c1-900; tail -1 $O/pinned_$S.txt
done
tail -3 $O/pytest_a.log; cat $O/kbench.log | cut -c1-300; cat $O/clock_under_load.txt | tail -15
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3))"; done
