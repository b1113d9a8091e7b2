#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2v
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 600 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
cut -c1-260 $O/bench_cfg5.json
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 0 > $O/prof.log 2>&1
echo "rocprof rc $?"
cd $GRAFT_REPO_ROOT
python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1
python tools/step_gaps.py $O/s5/b_kernel_trace.csv 2 > $O/gaps.txt 2>&1
find $O/s5 -name "*kernel_trace.csv" -delete
cat $O/breakdown.txt | head -60
