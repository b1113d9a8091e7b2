#!/bin/bash
# round 2, first GPU job: sanity tests, kernel micro-benchmarks, pretrain probe, stats for bf16x3 / f32
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 300 python tools/kbench.py edge chamfer knn > $O/kbench.log 2>&1
timeout 600 python tools/pretrain_probe.py > $O/probe.log 2>&1
for A in bf16x3 f32; do
  export PARSENET_MS_ARITH=$A
  timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_$A.json 2> $O/bench_cfg5_$A.err
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/s5_$A -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof_$A.log 2>&1)
  find $O/s5_$A -name "*kernel_trace.csv" -delete
done
tail -3 $O/pytest.log; cat $O/kbench.log; cat $O/probe.log; cat $O/bench_cfg5_*.json
