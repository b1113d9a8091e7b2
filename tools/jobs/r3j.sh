#!/bin/bash
# round 3, job j: edge-conv reduce with register-resident neighbour lists; auto mode on four samples
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
mkdir -p $O
timeout 900 python -m pytest tests/test_edgeconv_gpu.py tests/test_encoder_gpu.py tests/test_fused_gpu.py tests/test_golden_gpu.py tests/test_workloads_gpu.py tests/test_fitting_batch_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cfg5_b.json 2> $O/bench_cfg5_b.err
tail -2 $O/pytest.log
for f in bench_cfg4 bench_cfg5 bench_cfg5_b; do python3 -c "
import json
d=json.loads([l for l in open('$O/$f.json').read().splitlines() if l.startswith('{')][-1])
r=d['roofline'] or {}
k=d['kernels']
print('$f value %.2f ms %.2f dense %.2f'%(d['value'],d['ms_per_step'],d.get('value_dense') or 0), r.get('frac'), r.get('avg_launch_ms'), (r.get('meanshift_launches') or {}).get('timed_and_warmup_calls'), (r.get('block_sparse') or {}).get('tile_pairs_executed'), 'reduce_fwd', k.get('edgeconv_reduce_fwd'), 'wmax', k.get('weighted_max_fwd'))"; done
