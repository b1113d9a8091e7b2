#!/bin/bash
# round 3, job zg: column-pass ping-pong with and without the issue priorities, after the scalar-VALU change
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3zg
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pre_cfg5.pt
run_bench() { for i in 1 2; do timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > $O/bench_$1_$i.json 2> $O/bench_$1.err; done; }
run_bench prio
PN_EXTRA_HIPCC_FLAGS=-DX3_NOPRIO python -m parsenet_codebase_amd.build > $O/build_noprio.log 2>&1
run_bench noprio
PN_EXTRA_HIPCC_FLAGS="-DMS_TIMING -DX3_NOPRIO" python -m parsenet_codebase_amd.build > $O/build_t2.log 2>&1
PARSENET_MS_SPARSE=0 timeout 300 python tools/ms_timing.py 2>&1 | grep "PASS 2" | sed 's/^/noprio /'
PN_EXTRA_HIPCC_FLAGS="-DMS_TIMING" python -m parsenet_codebase_amd.build > $O/build_t1.log 2>&1
PARSENET_MS_SPARSE=0 timeout 300 python tools/ms_timing.py 2>&1 | grep "PASS 2" | sed 's/^/prio   /'
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3zg/bench_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); k=d["kernels"]
    print(f[-22:], "value %.2f ms/step %.2f fwd %.4f rows %.4f cols %.4f pairs %.3f"%(d["value"],d["ms_per_step"],k["meanshift_fwd"],k["meanshift_bwd_rows"],k["meanshift_bwd_cols"],d["roofline"]["block_sparse"]["tile_pairs_executed"]["mean"]))
P
