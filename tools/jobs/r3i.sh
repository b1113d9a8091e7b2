#!/bin/bash
# round 3, job i: ring allocated in one piece (cfg4 regression of the evidence run), order-preserving weighted-max backward
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence_r03
mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_fitting_batch_gpu.py tests/test_golden_gpu.py tests/test_e2e_gpu.py -m gpu -q > $O/pytest_i.log 2>&1; echo "rc $?" >> $O/pytest_i.log
timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_b.json 2> $O/bench_cfg4_b.err
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cfg5_c.json 2> $O/bench_cfg5_c.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cfg5_d.json 2> $O/bench_cfg5_d.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s4 -o b -- python3 $R/bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > $O/prof4.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o b -- python3 $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-dense --profile-steps 0 > $O/prof5.log 2>&1
cd $R
python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1
python tools/step_gaps.py $O/s5/b_kernel_trace.csv 4 > $O/gaps_final.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
tail -2 $O/pytest_i.log
for f in bench_cfg4 bench_cfg4_b bench_cfg5_c bench_cfg5_d; do python3 -c "
import json
d=json.loads([l for l in open('$O/$f.json').read().splitlines() if l.startswith('{')][-1])
r=d['roofline'] or {}
print('$f value %.2f ms %.2f dense %.2f'%(d['value'],d['ms_per_step'],d.get('value_dense') or 0), r.get('frac'), r.get('avg_launch_ms'), (r.get('meanshift_launches') or {}).get('timed_and_warmup_calls'), (r.get('block_sparse') or {}).get('tile_pairs_executed'))"; done
cat $O/breakdown.txt; grep "^step" $O/gaps_final.txt
