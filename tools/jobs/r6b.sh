#!/bin/bash
# round 6, call b: the weight-gradient kernel (tests, micro-benchmark), bench lines with the bf16 x 3 GEMM for every
# large product (new default) against the round-5 default (frozen weights only), alternating on one box; which
# whole-step parity tests move with the new rounding
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
timeout 900 python -m pytest tests/test_chamfer_gpu.py -m gpu -q > $O/pytest_chamfer.log 2>&1; echo "rc $?" >> $O/pytest_chamfer.log
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py tests/test_workloads_gpu.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "rc $?" >> $O/pytest_gemm.log
timeout 600 python tools/kbench.py chamfer wgrad gemm > $O/kbench.log 2>&1
export PARSENET_PRETRAIN_CACHE=/tmp/pre_x3.pt
for rep in 1 2; do
PARSENET_GEMM_X3=1 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_x3_$rep.json 2> $O/bench_cfg5_x3_$rep.err
PARSENET_PRETRAIN_CACHE=/tmp/pre_fr.pt PARSENET_GEMM_X3=frozen PARSENET_GEMM_X3_MIN_ROWS=512 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_frozen_$rep.json 2> $O/bench_cfg5_frozen_$rep.err
done
PARSENET_GEMM_X3=1 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_x3.json 2> $O/bench_cfg4_x3.err
PARSENET_GEMM_X3=frozen PARSENET_GEMM_X3_MIN_ROWS=512 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_frozen.json 2> $O/bench_cfg4_frozen.err
for W in cfg2 cfg3; do
PARSENET_GEMM_X3=1 timeout 600 python bench.py --workload $W --no-cpu-baseline > $O/bench_${W}_x3.json 2> $O/bench_${W}_x3.err
PARSENET_GEMM_X3=frozen PARSENET_GEMM_X3_MIN_ROWS=512 timeout 600 python bench.py --workload $W --no-cpu-baseline > $O/bench_${W}_frozen.json 2> $O/bench_${W}_frozen.err
done
unset PARSENET_PRETRAIN_CACHE
timeout 2400 python -m pytest tests/test_parity_fullsize_bwd_gpu.py tests/test_fullsize_gpu.py tests/test_e2e_gpu.py tests/test_trainer_gpu.py tests/test_determinism_gpu.py tests/test_golden_gpu.py tests/test_fitting_batch_gpu.py -m gpu -q -s > $O/pytest_whole.log 2>&1; echo "rc $?" >> $O/pytest_whole.log
tail -3 $O/pytest_chamfer.log; tail -3 $O/pytest_gemm.log; cat $O/kbench.log | cut -c1-260
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3))"; done
grep -n "passed\|failed\|FAILED\|^rc" $O/pytest_whole.log | head -20
