#!/bin/bash
# round 3, job zd: forward-pass ping-pong (PN_MS_PINGPONG=2) again, now with the scalar elementwise stage
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3zd
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pre_cfg5.pt
for pp in 2 1 2 1; do
  PN_MS_PINGPONG=$pp timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > $O/bench_pp${pp}_$RANDOM.json 2> $O/bench_pp$pp.err
done
PN_EXTRA_HIPCC_FLAGS=-DMS_TIMING python -m parsenet_codebase_amd.build > $O/build.log 2>&1
for pp in 2 1; do
PN_MS_PINGPONG=$pp PARSENET_MS_SPARSE=0 timeout 300 python tools/ms_timing.py 2>&1 | grep "PASS 0" | sed "s/^/pp$pp /"
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3zd/bench_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); k=d["kernels"]
    print(f[-22:], "value %.2f ms/step %.2f fwd %.4f rows %.4f cols %.4f pairs %.3f"%(d["value"],d["ms_per_step"],k["meanshift_fwd"],k["meanshift_bwd_rows"],k["meanshift_bwd_cols"],d["roofline"]["block_sparse"]["tile_pairs_executed"]["mean"]))
P
