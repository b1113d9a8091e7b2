"""GPU idle gaps from a rocprofv3 --kernel-trace CSV: python tools/gaps.py <kernel_trace.csv> [min_us] [last_ms]"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
if len(sys.argv) > 3:   # only the last <ms> of the trace (steady state)
    t0 = ev[-1][1] - float(sys.argv[3]) * 1e6
    ev = [e for e in ev if e[0] >= t0]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = Counter()
cnt = Counter()
tot = 0
cur_end = ev[0][1]
prev = ev[0][2]
for s, e, n in ev[1:]:
    g = s - cur_end
    if g > thr * 1000:
        key = (prev[:40], n[:40])
        gaps[key] += g
        cnt[key] += 1
        tot += g
    if e > cur_end:
        cur_end, prev = e, n
print("span %.1f ms, kernel busy %.1f ms, gaps > %.0f us: %.1f ms" % (span / 1e6, busy / 1e6, thr, tot / 1e6))
for k, v in gaps.most_common(25):
    print("%8.2f ms %5d x  %-40s -> %s" % (v / 1e6, cnt[k], k[0], k[1]))
