#!/bin/bash
# round 3, job ze: forward-pass ping-pong with the stage in two halves (second half opens H2), PN_MS_PINGPONG=2 vs 1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3ze
mkdir -p $O
PN_MS_PINGPONG=2 timeout 900 python -m pytest tests/test_meanshift_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
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
for f in sorted(glob.glob("gpurun_out/r3ze/bench_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); k=d["kernels"]
    print(f[-22:], "value %.2f ms/step %.2f fwd %.4f rows %.4f cols %.4f pairs %.3f"%(d["value"],d["ms_per_step"],k["meanshift_fwd"],k["meanshift_bwd_rows"],k["meanshift_bwd_cols"],d["roofline"]["block_sparse"]["tile_pairs_executed"]["mean"]))
P
