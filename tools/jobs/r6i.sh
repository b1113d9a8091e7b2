#!/bin/bash
# round 6, call i: standardisation of the spline segments as two launches: the suites that touch it, A/B bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6i; mkdir -p $O
timeout 1500 python -m pytest tests/test_fitting_batch_gpu.py tests/test_fitting_gpu.py tests/test_fitting_eval_gpu.py tests/test_golden_gpu.py tests/test_e2e_gpu.py tests/test_norms_gpu.py tests/test_determinism_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pre_i.pt
for rep in 1 2 3; do
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_$rep.json 2> $O/bench_cfg5_$rep.err
done
tail -6 $O/pytest.log
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['kernel_ms_per_step'].get('standardize'))"; done
