#!/bin/bash
# round 3, job zb: scalar residual subtractions in the bf16 x 3 split (default) against the packed form (-DX3_PACKED_VALU)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3zc
mkdir -p $O
timeout 900 python -m pytest tests/test_meanshift_gpu.py tests/test_knn_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pre_cfg5.pt
run_bench() { for i in 1 2; do timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > $O/bench_$1_$i.json 2> $O/bench_$1.err; done; }
run_bench scalar
PN_EXTRA_HIPCC_FLAGS=-DX3_PACKED_VALU python -m parsenet_codebase_amd.build > $O/build_packed.log 2>&1
run_bench packed
python -m parsenet_codebase_amd.build > $O/build_scalar.log 2>&1
run_bench scalar2
PN_EXTRA_HIPCC_FLAGS="-DMS_TIMING" python -m parsenet_codebase_amd.build > $O/build_t1.log 2>&1
PARSENET_MS_SPARSE=0 timeout 300 python tools/ms_timing.py 2>&1 | grep PASS | sed 's/^/scalar /'
PN_EXTRA_HIPCC_FLAGS="-DMS_TIMING -DX3_PACKED_VALU" python -m parsenet_codebase_amd.build > $O/build_t2.log 2>&1
PARSENET_MS_SPARSE=0 timeout 300 python tools/ms_timing.py 2>&1 | grep PASS | sed 's/^/packed /'
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3zc/bench_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); k=d["kernels"]
    print(f[-22:], "value %.2f ms/step %.2f fwd %.4f rows %.4f cols %.4f pairs %.3f knn_x3 %s"%(d["value"],d["ms_per_step"],k["meanshift_fwd"],k["meanshift_bwd_rows"],k["meanshift_bwd_cols"],d["roofline"]["block_sparse"]["tile_pairs_executed"]["mean"], {a:b for a,b in k.items() if "knn_x3" in a}))
P
