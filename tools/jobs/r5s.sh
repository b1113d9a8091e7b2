# k-means kernels of the locality order: tests, A/B of the cfg5 line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s
mkdir -p $O
timeout 1200 python -m pytest tests/test_meanshift_gpu.py tests/test_determinism_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for i in 1 2; do for KM in 1 0; do
PARSENET_MS_KMEANS_KERNELS=$KM timeout 600 python bench.py --no-cpu-baseline > $O/b${KM}_$i.json 2> $O/b${KM}_$i.err
python - <<PY
import json
d=json.load(open("$O/b${KM}_$i.json"))
print("kernels=$KM", round(d["value"],1), round(d["ms_per_step"],2), d["roofline"].get("frac"), d["config"].get("clusters_per_shape"), d["roofline"].get("tile_pairs_share", d["roofline"].get("block_sparse")))
PY
done; done
