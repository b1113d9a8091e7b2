#!/bin/bash
# round 3, job g: pinned staging ring — tests of the fitting stage, bench lines, gap analysis
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
mkdir -p $O
timeout 900 python -m pytest tests/test_fitting_batch_gpu.py tests/test_e2e_gpu.py tests/test_golden_gpu.py tests/test_trainer_gpu.py tests/test_fused_gpu.py tests/test_workloads_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err
timeout 900 python bench.py --no-cpu-baseline --steps 40 > $O/bench3.json 2> $O/bench3.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o b -- python3 $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-dense --profile-steps 0 > $O/prof5.log 2>&1
cd $R
python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1
python tools/step_gaps.py $O/s5/b_kernel_trace.csv 6 > $O/gaps.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
tail -3 $O/pytest.log
for f in $O/bench1.json $O/bench2.json $O/bench3.json; do python3 -c "
import json
d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1])
r=d['roofline'] or {}
print('value %.2f ms %.2f dense %.2f'%(d['value'],d['ms_per_step'],d.get('value_dense') or 0), r.get('frac'), r.get('avg_launch_ms'), (r.get('meanshift_launches') or {}).get('timed_and_warmup_calls'), (r.get('block_sparse') or {}).get('tile_pairs_executed'))"; done
cat $O/breakdown.txt; grep "^step\|^ *[0-9]*\.[0-9] us" $O/gaps.txt | awk '/^step/{print; n=0; next} n<3{print; n++}'
