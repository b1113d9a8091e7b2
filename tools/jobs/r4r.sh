cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4r
export PARSENET_PRETRAIN_CACHE=/tmp/pt_cache.pt
python bench.py --no-cpu-baseline --no-dense --steps 4 --warmup 1 > /dev/null 2>&1
for cfg in "1 384" "1 384" "1 384" "0 384" "0 384" "0 384" "1 0" "1 0" "1 0"; do
set -- $cfg
PARSENET_MS_NEAREST=$1 PARSENET_MS_FINE=$2 python bench.py --no-cpu-baseline --no-dense --steps 8 --warmup 2 > gpurun_out/r4r/b.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4r/b.json").read().splitlines() if l.startswith("{")][-1])
print("nearest $1 fine $2:", d["config"]["clusters_per_shape"], d["config"]["segments_per_shape"], d["roofline"]["passes"]["meanshift_bwd_cols"]["tile_pairs_executed"], round(d["value"],1))
PY
done
