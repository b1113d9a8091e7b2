#!/bin/bash
# round 6, call r: the covariance download of the spline standardisation queued before the primitives' fits
# (PARSENET_STD_EARLY existed only in the tree this call ran on: the change was measured and reverted, profiles/r06_host_overlap_ab.txt)
# (PARSENET_STD_EARLY); groups of shapes in the fitting stage re-measured (PARSENET_FIT_CHUNKS=2, last measured in round 3)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6r; mkdir -p $O
timeout 2400 python -m pytest tests/test_fitting_batch_gpu.py tests/test_golden_gpu.py tests/test_parity_fullsize_bwd_gpu.py tests/test_e2e_gpu.py tests/test_workloads_gpu.py tests/test_determinism_gpu.py -q -m gpu > $O/pytest_whole.log 2>&1; echo "rc $?" >> $O/pytest_whole.log
for i in 1 2; do
  PARSENET_STD_EARLY=1 timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_early_$i.json 2> $O/bench_cfg5_early_$i.err
  PARSENET_STD_EARLY=0 timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_late_$i.json 2> $O/bench_cfg5_late_$i.err
done
PARSENET_FIT_CHUNKS=2 timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_chunks2.json 2> $O/bench_cfg5_chunks2.err
tail -4 $O/pytest_whole.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6r/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
