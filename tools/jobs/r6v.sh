#!/bin/bash
# round 6, call v: workgroups of a planned mean-shift launch (PN_MS_FLAT_G: 256 = one per CU, the default; 512 / 1024 = the
# dispatcher hands out the later ones as CUs free up — does it cut the launch's tail?), bench.py cfg5 alternating on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6v; mkdir -p $O
for i in 1 2; do
  for G in 256 512 1024; do
    PN_MS_FLAT_G=$G timeout 900 python bench.py --no-cpu-baseline --no-dense > $O/bench_cfg5_g${G}_$i.json 2> $O/bench_cfg5_g${G}_$i.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6v/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
        print(f.split("/")[-1], round(d["value"], 1), round(d["ms_per_step"], 3), "fwd launch ms", r.get("avg_launch_ms"), "frac", r.get("frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
