#!/bin/bash
# round 3, job a: sanity of the GPU suite after the bench/workload refactor, held-out pre-training
# probe, first cfg2/cfg3 bench lines + rocprof kernel statistics
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
mkdir -p $O
timeout 900 python tools/pretrain_probe.py 1e-2 64 > $O/probe_lr1e-2.log 2>&1
timeout 600 python tools/pretrain_probe.py 3e-3 64 > $O/probe_lr3e-3.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
for W in cfg2 cfg3; do
timeout 600 python bench.py --workload $W --steps 20 --warmup 5 > $O/bench_$W.json 2> $O/bench_$W.err
done
cd /tmp
for W in cfg2 cfg3; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$W -o b -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > $O/prof_$W.log 2>&1
find $O/s_$W -name "*kernel_trace.csv" -delete
done
cd $R
grep -v "amdgpu.ids" $O/probe_lr1e-2.log | tail -20; grep -v "amdgpu.ids" $O/probe_lr3e-3.log | tail -20
tail -5 $O/pytest.log; for W in cfg2 cfg3; do cut -c1-600 $O/bench_$W.json; tail -3 $O/bench_$W.err; done
