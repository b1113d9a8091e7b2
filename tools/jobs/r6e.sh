#!/bin/bash
# round 6, call e: three-product threshold pass of the kNN engine, transposed graph prefetched on a side stream,
# FlatAdam in the trainers, whole-step tests on the pinned pre-training recipe: suites of the touched families,
# A/B bench lines on one box (alternating)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
timeout 2400 python -m pytest tests/test_knn_gpu.py tests/test_edgeconv_gpu.py tests/test_fullsize_gpu.py tests/test_meanshift_gpu.py tests/test_chamfer_gpu.py tests/test_trainer_gpu.py tests/test_fused_gpu.py tests/test_determinism_gpu.py tests/test_parity_fullsize_bwd_gpu.py tests/test_e2e_gpu.py -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python tools/kbench.py knn chamfer > $O/kbench.log 2>&1
export PARSENET_PRETRAIN_CACHE=/tmp/pre_e.pt
for rep in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_new_$rep.json 2> $O/bench_cfg5_new_$rep.err
PN_KNN_X3_P1=6 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_p1six_$rep.json 2> $O/bench_cfg5_p1six_$rep.err
PARSENET_CSR_PREFETCH=0 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_nocsr_$rep.json 2> $O/bench_cfg5_nocsr_$rep.err
done
for rep in 1 2; do
timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_new_$rep.json 2> $O/bench_cfg4_new_$rep.err
PN_KNN_X3_P1=6 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_p1six_$rep.json 2> $O/bench_cfg4_p1six_$rep.err
PARSENET_CSR_PREFETCH=0 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_nocsr_$rep.json 2> $O/bench_cfg4_nocsr_$rep.err
done
tail -30 $O/pytest.log; cut -c1-250 $O/kbench.log
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3), {k:v for k,v in d['kernel_ms_per_step'].items() if 'knn_x3' in k or 'sel_x3' in k or 'csr' in k or 'knn_final' in k})"; done
