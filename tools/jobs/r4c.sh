set -x
python bench.py --no-cpu-baseline > gpurun_out/r4c_cfg5_a.json 2> gpurun_out/r4c_cfg5_a.err
python bench.py --no-cpu-baseline > gpurun_out/r4c_cfg5_b.json 2> gpurun_out/r4c_cfg5_b.err
python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/r4c_cfg4.json 2>/dev/null
python bench.py --workload cfg3 --no-cpu-baseline > gpurun_out/r4c_cfg3.json 2>/dev/null
python bench.py --workload cfg2 --no-cpu-baseline > gpurun_out/r4c_cfg2.json 2>/dev/null
python - <<'PY'
import json
for f in ("cfg5_a","cfg5_b","cfg4","cfg3","cfg2"):
    try:
        d=json.loads(open("gpurun_out/r4c_%s.json"%f).read().strip().splitlines()[-1])
        r=d.get("roofline") or {}
        print(f, round(d["value"],2), round(d["ms_per_step"],2), d.get("value_dense"), d["config"].get("pretrain_final_loss"), d["config"].get("clusters_per_shape"), (r.get("block_sparse") or {}).get("tile_pairs_executed"), r.get("frac"))
        print("   ", {k:v for k,v in d["kernels"].items() if "edgeconv" in k or "triplet" in k or "chamfer" in k or "gather" in k})
    except Exception as e:
        print(f, "ERR", e)
PY
