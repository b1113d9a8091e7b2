#!/bin/bash
# round 6, call k: scalar Chamfer kernel with ONE candidate range per query (no merge / memset / unpack launches) once the
# query workgroups alone fill the chip: micro-benchmark at the thresholds 384 (new default) / 0 (round 5's splitting),
# the suite, cfg2 / cfg3 bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6k; mkdir -p $O
timeout 300 python tools/kbench.py chamfer > $O/kbench_direct384.log 2>&1
PN_CHAMFER_DIRECT_WGS=0 timeout 300 python tools/kbench.py chamfer > $O/kbench_split.log 2>&1
PN_CHAMFER_DIRECT_WGS=128 timeout 300 python tools/kbench.py chamfer > $O/kbench_direct128.log 2>&1
timeout 600 python -m pytest tests/test_chamfer_gpu.py tests/test_fitting_batch_gpu.py tests/test_workloads_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
for rep in 1 2; do
timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2_direct_$rep.json 2> $O/bench_cfg2_direct_$rep.err
PN_CHAMFER_DIRECT_WGS=0 timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2_split_$rep.json 2> $O/bench_cfg2_split_$rep.err
done
grep "scalar" $O/kbench_direct384.log | cut -c1-150; echo; grep "scalar" $O/kbench_split.log | cut -c1-150; echo; grep "scalar" $O/kbench_direct128.log | cut -c1-150
tail -3 $O/pytest.log
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(round(d['value'],1), round(d['ms_per_step'],3))"; done
