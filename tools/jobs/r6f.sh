#!/bin/bash
# round 6, call f: approximate kNN passes on CENTRED rows (+ three-product threshold pass for the graphs): bit-exact
# suites, the randomised sweep, A/B bench lines against PN_KNN_X3_CENTRE=0 on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
timeout 600 python tools/probes/copy_sites.py 8 > $O/copy_sites.txt 2>&1
timeout 1800 python -m pytest tests/test_knn_gpu.py tests/test_fullsize_gpu.py tests/test_golden_gpu.py tests/test_encoder_gpu.py tests/test_edgeconv_gpu.py tests/test_determinism_gpu.py tests/test_workloads_gpu.py -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 500 python tools/fuzz.py 300 > $O/fuzz.log 2>&1
timeout 300 python tools/kbench.py knn > $O/kbench_centre.log 2>&1
PN_KNN_X3_CENTRE=0 timeout 300 python tools/kbench.py knn > $O/kbench_nocentre.log 2>&1
export PARSENET_PRETRAIN_CACHE=/tmp/pre_f.pt
for rep in 1 2; do
timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_centre_$rep.json 2> $O/bench_cfg4_centre_$rep.err
PN_KNN_X3_CENTRE=0 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_nocentre_$rep.json 2> $O/bench_cfg4_nocentre_$rep.err
PN_KNN_X3_P1=6 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_centre6_$rep.json 2> $O/bench_cfg4_centre6_$rep.err
done
for rep in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_centre_$rep.json 2> $O/bench_cfg5_centre_$rep.err
PN_KNN_X3_CENTRE=0 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_nocentre_$rep.json 2> $O/bench_cfg5_nocentre_$rep.err
done
head -70 $O/copy_sites.txt | cut -c1-330; tail -12 $O/pytest.log; tail -6 $O/fuzz.log | cut -c1-300; head -4 $O/kbench_centre.log; head -4 $O/kbench_nocentre.log
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3), {k:v for k,v in d['kernel_ms_per_step'].items() if 'knn_x3' in k or 'knn_final' in k or 'fallback' in k or 'scan' in k})"; done
