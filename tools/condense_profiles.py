"""Condense the output of tools/prof_round.sh (gpurun_out/r1b) into profiles/: python tools/condense_profiles.py"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "r1b")
P = os.path.join(ROOT, "profiles")


def condense(src, dst, steps=12):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "CallsPerStep", "TotalMs", "MsPerStep", "AvgUs", "Percentage"])
        for r in rows[:50]:
            c, t = int(r["Calls"]), float(r["TotalDurationNs"])
            w.writerow([r["Name"][:140], c, round(c / steps, 2), round(t / 1e6, 3), round(t / 1e6 / steps, 4),
                        round(t / c / 1e3, 2), round(100 * t / tot, 2)])
    return tot / 1e6 / steps


PMC_ONLY = "--pmc-only" in sys.argv   # used on the GPU box between the counter and the bench runs
for n in (() if PMC_ONLY else ("cfg5", "cfg4")):
    line = open(os.path.join(R, "bench_%s.json" % n)).read().strip()
    d = json.loads(line)
    open(os.path.join(P, "r01_bench_%s.json" % n), "w").write(line + "\n")
    print(n, round(d["value"], 1), "shapes/s", round(d["ms_per_step"], 2), "ms/step", d["roofline"]["kernel"],
          round(d["roofline"]["frac"], 3), d["roofline"]["traffic"])
if not PMC_ONLY:
    print("cfg5 kernel ms/step", round(condense(os.path.join(R, "s5", "b_kernel_stats.csv"),
                                                os.path.join(P, "r01_cfg5_kernel_stats.csv")), 2))
    print("cfg4 kernel ms/step", round(condense(os.path.join(R, "s4", "b_kernel_stats.csv"),
                                                os.path.join(P, "r01_cfg4_kernel_stats.csv")), 2))
# mean-shift kernel family of the counter runs: the default arithmetic (fp16 x 2)
KERNEL, PMC_FILE = "pn_msh_kernel", "r01_meanshift_h2_pmc.csv"
agg = collections.OrderedDict()
for n in (1, 2, 3, 4):
    for r in csv.DictReader(open(os.path.join(R, "pmc%d" % n, "p_counter_collection.csv"))):
        k = r["Kernel_Name"]
        if KERNEL not in k:
            continue
        a = agg.setdefault((k.split("(")[0].replace("void ", ""), r["Counter_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
with open(os.path.join(P, PMC_FILE), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "avg_per_launch"])
    for (k, c), (n, v) in sorted(agg.items()):
        w.writerow([k, c, n, round(v / n, 4)])
g = lambda k, c: agg[(k, c)][1] / agg[(k, c)][0]   # noqa: E731
for p_ in (0, 1, 2):
    k = "%s<%d>" % (KERNEL, p_)
    print(k, "mfma busy/(4*wave cycles)", round(g(k, "SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g(k, "SQ_WAVE_CYCLES")), 3),
          "FETCH MB", round(g(k, "FETCH_SIZE") / 1024, 1), "WRITE MB", round(g(k, "WRITE_SIZE") / 1024, 1),
          "LDS conflicts", g(k, "SQ_LDS_BANK_CONFLICT"))
