#!/bin/bash
# round 3, job d: pipelined fitting stage (tests + how much it buys), run-to-run variance of the cfg5 line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_fitting_batch_gpu.py tests/test_e2e_gpu.py tests/test_golden_gpu.py -m gpu -q -s --durations=5 > $O/pytest_a.log 2>&1; echo "rc $?" >> $O/pytest_a.log
timeout 900 python -m pytest tests/test_parity_fullsize_bwd_gpu.py -m gpu -q -s -k "whole_e2e" > $O/pytest_b.log 2>&1; echo "rc $?" >> $O/pytest_b.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/first.json 2> $O/first.err
for rep in 1 2 3; do for C in 1 2 4; do
PARSENET_FIT_CHUNKS=$C timeout 600 python bench.py --no-cpu-baseline --no-dense --profile-steps 0 > $O/chunks${C}_$rep.json 2> $O/chunks${C}_$rep.err
done; done
cd /tmp
for C in 1 2; do
PARSENET_FIT_CHUNKS=$C timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5_c$C -o b -- python3 $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-dense --profile-steps 0 > $O/prof5_c$C.log 2>&1
python3 $R/tools/step_breakdown.py $O/s5_c$C/b_kernel_trace.csv > $O/breakdown_c$C.txt 2>&1
done
cd $R
find $O -name "*kernel_trace.csv" -delete
grep -i "parity\|passed\|failed\|^rc \|Error" $O/pytest_a.log $O/pytest_b.log | cut -c1-1000
for f in $O/first.json $O/chunks*.json; do echo $f; python3 -c "
import json,sys
d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1])
print('  value %.2f ms %.2f'%(d['value'],d['ms_per_step']), d['config'].get('pretrain_final_loss'))"; done
cat $O/breakdown_c1.txt $O/breakdown_c2.txt
