#!/bin/bash
# round 6, call j: same-box A/B of the late changes: standardisation's fused selection / extents (PARSENET_STD_FUSED),
# and the whole round against an emulation of round 5's defaults
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pre_j.pt
for rep in 1 2 3; do
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_new_$rep.json 2> $O/bench_cfg5_new_$rep.err
PARSENET_STD_FUSED=0 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_stdoff_$rep.json 2> $O/bench_cfg5_stdoff_$rep.err
PARSENET_PRETRAIN_CACHE=/tmp/pre_j5.pt PARSENET_STD_FUSED=0 PN_KNN_X3_CENTRE=0 PN_KNN_X3_P1=6 PARSENET_FLAT_ADAM=0 PN_CHAMFER_MFMA=0 timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_r5like_$rep.json 2> $O/bench_cfg5_r5like_$rep.err
done
for rep in 1 2; do
timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_new_$rep.json 2> $O/bench_cfg4_new_$rep.err
PN_KNN_X3_CENTRE=0 PN_KNN_X3_P1=6 PARSENET_FLAT_ADAM=0 timeout 600 python bench.py --workload cfg4 --steps 30 --no-cpu-baseline > $O/bench_cfg4_r5like_$rep.json 2> $O/bench_cfg4_r5like_$rep.err
done
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3))"; done
