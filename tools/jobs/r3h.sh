#!/bin/bash
# round 3, job h: with the pinned ring in place: pipelined groups again, and repeats for the variance
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --no-cpu-baseline --no-dense --profile-steps 0 > $O/first.json 2> $O/first.err
for rep in 1 2; do for C in 1 2 4; do
PARSENET_FIT_CHUNKS=$C timeout 600 python bench.py --no-cpu-baseline --no-dense --profile-steps 0 --steps 40 > $O/chunks${C}_$rep.json 2> $O/chunks${C}_$rep.err
done; done
for f in $O/first.json $O/chunks*.json; do echo -n "$f "; python3 -c "
import json
d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1])
print('value %.2f ms %.2f'%(d['value'],d['ms_per_step']))"; done
