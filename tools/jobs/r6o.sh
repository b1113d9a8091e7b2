#!/bin/bash
# round 6, call o: the launch diet (conv weights as views, _EdgeWeight, unbind of the per-shape losses, one-launch
# locality order, one-launch tails of the max variant): suites that touch it, operators by line, same-box A/B against
# the tree before it (exported to _ab_prev/ for this call only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
timeout 1500 python -m pytest tests/test_meanshift_gpu.py tests/test_norms_gpu.py tests/test_edgeconv_gpu.py tests/test_host_logic.py -q -x -m "gpu or not gpu" -k "not world" > $O/pytest_units.log 2>&1; echo "rc $?" >> $O/pytest_units.log
timeout 1800 python -m pytest tests/test_golden_gpu.py tests/test_parity_fullsize_bwd_gpu.py tests/test_e2e_gpu.py tests/test_encoder_gpu.py tests/test_workloads_gpu.py tests/test_determinism_gpu.py -q -m gpu > $O/pytest_whole.log 2>&1; echo "rc $?" >> $O/pytest_whole.log
timeout 600 python tools/probes/op_lines.py > $O/op_lines.txt 2> $O/op_lines.err
for i in 1 2; do
  timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_new_$i.json 2> $O/bench_cfg5_new_$i.err
  if [ -d _ab_prev ]; then
    (cd _ab_prev && timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > ../$O/bench_cfg5_prev_$i.json 2> ../$O/bench_cfg5_prev_$i.err)
  fi
done
timeout 900 python bench.py --workload cfg4 --no-cpu-baseline > $O/bench_cfg4_new.json 2> $O/bench_cfg4_new.err
tail -3 $O/pytest_units.log; tail -4 $O/pytest_whole.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6o/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
head -3 $O/op_lines.txt
