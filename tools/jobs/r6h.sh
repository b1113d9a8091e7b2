#!/bin/bash
# round 6, call h: collecting pass of the centred kNN graphs on three piece products: bit-exact suites, the randomised
# sweep, A/B against PN_KNN_X3_P2=6 and against the round-5 form (PN_KNN_X3_CENTRE=0) on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
timeout 1500 python -m pytest tests/test_knn_gpu.py tests/test_fullsize_gpu.py tests/test_golden_gpu.py tests/test_encoder_gpu.py tests/test_determinism_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 400 python tools/fuzz.py 240 > $O/fuzz.log 2>&1
timeout 300 python tools/kbench.py knn > $O/kbench_p2_3.log 2>&1
PN_KNN_X3_P2=6 timeout 300 python tools/kbench.py knn > $O/kbench_p2_6.log 2>&1
export PARSENET_PRETRAIN_CACHE=/tmp/pre_h.pt
for rep in 1 2; do
timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_p2three_$rep.json 2> $O/bench_cfg4_p2three_$rep.err
PN_KNN_X3_P2=6 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_p2six_$rep.json 2> $O/bench_cfg4_p2six_$rep.err
PN_KNN_X3_CENTRE=0 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_r5form_$rep.json 2> $O/bench_cfg4_r5form_$rep.err
done
for rep in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_p2three_$rep.json 2> $O/bench_cfg5_p2three_$rep.err
PN_KNN_X3_P2=6 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_p2six_$rep.json 2> $O/bench_cfg5_p2six_$rep.err
done
tail -4 $O/pytest.log; tail -2 $O/fuzz.log | cut -c1-300; head -2 $O/kbench_p2_3.log | tail -1; head -2 $O/kbench_p2_6.log | tail -1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3), {k:v for k,v in d['kernel_ms_per_step'].items() if 'knn_x3' in k or 'knn_final' in k or 'fallback' in k or 'scan' in k})"; done
