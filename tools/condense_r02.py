"""Condense rocprofv3 outputs under gpurun_out/ into the committed summaries of profiles/ (round 2).
python tools/condense_r02.py named gpurun_out/r2g      -> r02_named_kernels_{stats,pmc}.csv
python tools/condense_r02.py bench gpurun_out/<dir>/s5_<arith> <tag> <steps>  -> r02_cfg5_<tag>_kernel_stats.csv"""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
NAMED = ("pn_edge_feature", "pn_edgeconv_reduce", "pn_chamfer_nn")


def stats(src, dst, steps, keep=60):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "CallsPerStep", "TotalMs", "MsPerStep", "AvgUs", "MinUs", "MaxUs", "Percentage"])
        for r in rows[:keep]:
            c, t = int(r["Calls"]), float(r["TotalDurationNs"])
            w.writerow([r["Name"][:140], c, round(c / steps, 2), round(t / 1e6, 3), round(t / 1e6 / steps, 4),
                        round(t / c / 1e3, 2), round(float(r["MinNs"]) / 1e3, 2), round(float(r["MaxNs"]) / 1e3, 2),
                        round(100 * t / tot, 2)])
    return tot / 1e6 / steps


def pmc(dirs, dst, names):
    agg = collections.OrderedDict()
    for d in dirs:
        fn = os.path.join(d, "p_counter_collection.csv")
        if not os.path.exists(fn):
            continue
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if not any(t in k for t in names):
                continue
            key = (k.split("(")[0].replace("void ", ""), r.get("Grid_Size", ""), r["Counter_Name"])
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_size", "counter", "launches", "avg_per_launch"])
        for (k, g, c), (n, v) in agg.items():
            w.writerow([k, g, c, n, round(v / n, 4)])


if sys.argv[1] == "named":
    src = sys.argv[2]
    rows = [r for r in csv.DictReader(open(os.path.join(src, "kstats", "k_kernel_stats.csv")))
            if any(t in r["Name"] for t in NAMED)]
    with open(os.path.join(P, "r02_named_kernels_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "AvgUs", "MinUs", "MaxUs"])
        for r in rows:
            w.writerow([r["Name"][:140], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2),
                        round(float(r["MinNs"]) / 1e3, 2), round(float(r["MaxNs"]) / 1e3, 2)])
    pmc([os.path.join(src, "pmc%d" % i) for i in range(1, 9)], os.path.join(P, "r02_named_kernels_pmc.csv"), NAMED)
elif sys.argv[1] == "bench":
    print(stats(os.path.join(sys.argv[2], "b_kernel_stats.csv"),
                os.path.join(P, "r02_cfg5_%s_kernel_stats.csv" % sys.argv[3]), int(sys.argv[4])))
elif sys.argv[1] == "pmc":
    pmc(sys.argv[4:], os.path.join(P, sys.argv[2]), tuple(sys.argv[3].split(",")))
