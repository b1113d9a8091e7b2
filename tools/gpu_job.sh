#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -12 $O/pytest.log
for i in 1 2; do
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg5_$i.json 2> $O/bench_cfg5_$i.err; cut -c1-200 $O/bench_cfg5_$i.json
done
python - <<'P'
import json
d = json.load(open("gpurun_out/r3r/bench_cfg5_2.json")); k = d["kernels"]
print({n: k[n] for n in k if n.startswith("sel") or n.startswith("knn")})
P
