#!/bin/bash
# round 3, job e: whole GPU suite on the current tree, default bench lines, the SplineNet hipGraph switch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -s --durations=6 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err
for rep in 1 2; do for G in 0 1; do
PARSENET_SPLINE_GRAPH=$G timeout 600 python bench.py --no-cpu-baseline --no-dense --profile-steps 0 > $O/graph${G}_$rep.json 2> $O/graph${G}_$rep.err
done; done
cd /tmp
for G in 0 1; do
PARSENET_SPLINE_GRAPH=$G timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5_g$G -o b -- python3 $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-dense --profile-steps 0 > $O/prof5_g$G.log 2>&1
python3 $R/tools/step_breakdown.py $O/s5_g$G/b_kernel_trace.csv > $O/breakdown_g$G.txt 2>&1
done
cd $R
find $O -name "*kernel_trace.csv" -delete
grep -i "parity\|eval-mode\|passed\|failed\|^rc \|Error" $O/pytest.log | cut -c1-900
for f in $O/bench1.json $O/graph*.json; do echo $f; python3 -c "
import json
d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1])
r=d['roofline'] or {}
print('  value %.2f ms %.2f dense %.2f'%(d['value'],d['ms_per_step'],d.get('value_dense') or 0), r.get('frac'), r.get('avg_launch_ms'), (r.get('meanshift_launches') or {}).get('timed_and_warmup_calls'), (r.get('block_sparse') or {}).get('tile_pairs_executed'))"; tail -2 ${f%.json}.err | grep -v amdgpu; done
cat $O/breakdown_g0.txt $O/breakdown_g1.txt
