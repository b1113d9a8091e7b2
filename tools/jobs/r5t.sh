# locality order: Lloyd steps and fine cells against the share of tile pairs the plans keep and the step time
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5t
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r5t.pt
for cfg in "2 384" "3 384" "4 384" "2 512" "3 512" "2 384"; do
set -- $cfg
PARSENET_MS_LLOYD=$1 PARSENET_MS_FINE=$2 timeout 600 python bench.py --no-cpu-baseline > $O/b_$1_$2.json 2> $O/b_$1_$2.err
python - <<PY
import json
d=json.load(open("$O/b_$1_$2.json"))
r=d["roofline"]
print("lloyd=$1 fine=$2", round(d["value"],1), round(d["ms_per_step"],2), "pairs", round(r["block_sparse"]["tile_pairs_executed"]["mean"],4), "fwd ms", round(r["avg_launch_ms"],4))
PY
done
