"""Cross-check of bench.py's counter-backed roofline against rocprofv3 (round 4, verdict item 2).

Inputs: the JSON line a ``bench.py --profile-only`` process printed UNDER rocprofv3, the kernel
statistics of that same process (``--kernel-trace --stats``) and the counter collection of a second
``--profile-only`` process (``--pmc SQ_VALU_MFMA_BUSY_CYCLES``).  For every mean-shift pass:
  frac_pmc   = SQ_VALU_MFMA_BUSY_CYCLES / 32 x 32768 FLOP / AvgNs / 2.5e15   (rocprofv3 only)
  frac_bench = roofline.passes[...].frac of the line            (in-kernel counters / HIP events)
  python tools/roofline_check.py <line.json> <kernel_stats.csv> <counter_collection.csv> [out.txt]
"""
import collections
import csv
import json
import sys

PASS = {"pn_ms3_kernel<0": "meanshift_fwd", "pn_ms3_kernel<1": "meanshift_bwd_rows", "pn_ms3_kernel<2": "meanshift_bwd_cols"}


def main():
    line = [ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1]
    roof = json.loads(line)["roofline"]
    avg_ns, calls = {}, {}
    for r in csv.DictReader(open(sys.argv[2])):
        for k, fam in PASS.items():
            if k in r["Name"]:
                avg_ns[fam] = float(r["TotalDurationNs"]) / int(r["Calls"])
                calls[fam] = int(r["Calls"])
    cyc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(sys.argv[3])):
        if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        for k, fam in PASS.items():
            if k in r["Kernel_Name"]:
                cyc[fam][0] += 1
                cyc[fam][1] += float(r["Counter_Value"])
    out = ["# pass: launches (stats / pmc run), AvgUs (rocprofv3), SQ_VALU_MFMA_BUSY_CYCLES per launch, GFLOP per launch "
           "from the counter, from bench.py's in-kernel pair count; fraction of 2.5 PFLOP/s: rocprofv3-only / bench line"]
    worst = 0.0
    for fam in ("meanshift_fwd", "meanshift_bwd_rows", "meanshift_bwd_cols"):
        if fam not in avg_ns or fam not in cyc or fam not in roof.get("passes", {}):
            continue
        n, tot = cyc[fam]
        gflop_pmc = tot / n / 32.0 * 32768.0 / 1e9
        p = roof["passes"][fam]
        units = {"meanshift_fwd": 2, "meanshift_bwd_rows": 3, "meanshift_bwd_cols": 4}[fam]
        gflop_cnt = p["tile_pairs_executed"] / p["launches"] * 2.0 * 32 * 32 * 128 * units * 6 / 1e9
        frac_pmc = gflop_pmc * 1e9 / (avg_ns[fam] * 1e-9) / 2.5e15
        dev = abs(frac_pmc - p["frac"]) / p["frac"]
        worst = max(worst, dev)
        out.append("%-20s launches %d / %d  AvgUs %.2f (bench events %.2f)  cycles %.4e  GFLOP %.1f (counter) %.1f (pairs)  "
                   "frac %.4f (rocprofv3) %.4f (bench line)  deviation %.2f %%"
                   % (fam, calls[fam], n, avg_ns[fam] / 1e3, p["avg_launch_ms"] * 1e3, tot / n, gflop_pmc, gflop_cnt,
                      frac_pmc, p["frac"], 100 * dev))
    out.append("largest deviation between the bench line and the rocprofv3-only figure: %.2f %%" % (100 * worst))
    txt = "\n".join(out) + "\n"
    sys.stdout.write(txt)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt)


if __name__ == "__main__":
    main()
