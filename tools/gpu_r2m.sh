#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2m
mkdir -p $O
timeout 900 python -m pytest tests/test_meanshift_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/b_$tag.json 2>$O/b_$tag.err; python - <<PY
import json
d=json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1])
print("$tag", round(d["value"],2), round(d["ms_per_step"],2), {k:v for k,v in d["kernels"].items() if k.startswith("meanshift")})
PY
}
run dense PARSENET_MS_SPARSE=0
run lloyd0 PARSENET_MS_LLOYD=0
run lloyd2 PARSENET_MS_LLOYD=2
run lloyd4 PARSENET_MS_LLOYD=4
run lloyd2_s66 PARSENET_MS_LLOYD=2 PN_MS_SLICES=6,6
run lloyd2_s48 PARSENET_MS_LLOYD=2 PN_MS_SLICES=4,8
run lloyd2_s22 PARSENET_MS_LLOYD=2 PN_MS_SLICES=2,2
tail -4 $O/pytest.log
