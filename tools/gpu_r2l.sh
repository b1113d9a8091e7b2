#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
timeout 900 python -m pytest tests/test_meanshift_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/b_sparse.json 2>$O/b_sparse.err
PARSENET_MS_SPARSE=0 timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/b_dense.json 2>$O/b_dense.err
tail -25 $O/pytest.log; cut -c1-300 $O/b_sparse.json; tail -3 $O/b_sparse.err; cut -c1-300 $O/b_dense.json
python - <<'PY'
import json
for f in ("b_sparse","b_dense"):
    try:
        d=json.loads(open("gpurun_out/r2l/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], {k:v for k,v in d["kernels"].items() if k.startswith("meanshift")})
    except Exception as e: print(f, e)
PY
