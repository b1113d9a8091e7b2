#!/bin/bash
# round 3, job f: suite on the current tree, bench lines, where the device waits for the host (gaps)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -s --durations=6 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o b -- python3 $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-dense --profile-steps 0 > $O/prof5.log 2>&1
cd $R
python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1
python tools/step_gaps.py $O/s5/b_kernel_trace.csv 3 > $O/gaps.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
timeout 600 python tools/host_cprofile.py > $O/host_cprofile.txt 2>&1
grep -i "parity\|eval-mode\|passed\|failed\|^rc \|Error" $O/pytest.log | cut -c1-700
for f in $O/bench1.json $O/bench2.json; do python3 -c "
import json
d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1])
r=d['roofline'] or {}
print('value %.2f ms %.2f dense %.2f'%(d['value'],d['ms_per_step'],d.get('value_dense') or 0), r.get('frac'), r.get('avg_launch_ms'), (r.get('meanshift_launches') or {}).get('timed_and_warmup_calls'), (r.get('block_sparse') or {}).get('tile_pairs_executed'))"; done
cat $O/breakdown.txt; cat $O/gaps.txt | head -60
