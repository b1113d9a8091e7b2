#!/bin/bash
# round 3, job k: kNN collecting pass with hit masks, edge-conv backward gather with register lists: parity + fuzz + bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
mkdir -p $O
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_fullsize_gpu.py tests/test_edgeconv_gpu.py tests/test_encoder_gpu.py tests/test_fused_gpu.py tests/test_golden_gpu.py tests/test_workloads_gpu.py tests/test_meanshift_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 300 python tools/fuzz.py 90 > $O/fuzz.log 2>&1; echo "rc $?" >> $O/fuzz.log
timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --workload cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
tail -2 $O/pytest.log; tail -3 $O/fuzz.log
for f in bench_cfg4 bench_cfg5 bench_cfg2 bench_cfg3; do python3 -c "
import json
d=json.loads([l for l in open('$O/$f.json').read().splitlines() if l.startswith('{')][-1])
r=d['roofline'] or {}
k=d['kernels']
print('$f value %.2f ms %.2f dense %.2f'%(d['value'],d['ms_per_step'],d.get('value_dense') or 0), r.get('frac'), (r.get('block_sparse') or {}).get('tile_pairs_executed'), 'reduce_fwd', k.get('edgeconv_reduce_fwd'), 'ec_bwd', k.get('edgeconv_bwd'), 'x3 pass2 c64', k.get('knn_x3_pass2_c64'), 'pass1', k.get('knn_x3_pass1_c64'))"; done
