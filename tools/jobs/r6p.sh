#!/bin/bash
# round 6, call p: one normalisation of the embedding per step (losses.normalized_rows): whole-step suites, A/B bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6p; mkdir -p $O
timeout 2400 python -m pytest tests/test_golden_gpu.py tests/test_parity_fullsize_bwd_gpu.py tests/test_e2e_gpu.py tests/test_workloads_gpu.py tests/test_determinism_gpu.py tests/test_trainer_gpu.py tests/test_fitting_batch_gpu.py tests/test_fitting_gpu.py tests/test_rccl_world1_gpu.py -q -m gpu > $O/pytest_whole.log 2>&1; echo "rc $?" >> $O/pytest_whole.log
for i in 1 2; do
  PARSENET_SHARE_NORMALIZE=1 timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_share_$i.json 2> $O/bench_cfg5_share_$i.err
  PARSENET_SHARE_NORMALIZE=0 timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_noshare_$i.json 2> $O/bench_cfg5_noshare_$i.err
done
tail -4 $O/pytest_whole.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6p/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
