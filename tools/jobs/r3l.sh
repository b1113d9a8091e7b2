#!/bin/bash
# round 3, job l: what the plan's relative threshold buys on the benchmark's embedding (same cached network)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/first.json 2> $O/first.err
for E in 1e-9 3e-8 1e-7; do
PARSENET_MS_SPARSE=1 PARSENET_MS_REL_EPS=$E timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/eps_$E.json 2> $O/eps_$E.err
done
for f in first eps_1e-9 eps_3e-8 eps_1e-7; do python3 -c "
import json
d=json.loads([l for l in open('$O/$f.json').read().splitlines() if l.startswith('{')][-1])
r=d['roofline'] or {}
k=d['kernels']
print('$f value %.2f ms %.2f'%(d['value'],d['ms_per_step']), r.get('frac'), (r.get('block_sparse') or {}).get('tile_pairs_executed'), 'ms kernels', k.get('meanshift_fwd'), k.get('meanshift_bwd_rows'), k.get('meanshift_bwd_cols'))"; done
