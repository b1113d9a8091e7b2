"""Largest GPU idle gaps inside the last steps of a rocprofv3 kernel trace of bench.py:
python tools/step_gaps.py trace.csv [steps] — per step: the gaps above 40 us with the kernel that
ended before and the one that started after (where the host made the GPU wait)."""
import csv
import sys
import collections

fn = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "multi_tensor_apply" in n or "pn_adam" in n]
bursts = []
for i in adam:
    if not bursts or i - bursts[-1][-1] > 50:
        bursts.append([i])
    else:
        bursts[-1].append(i)
ends = [b[-1] for b in bursts][-(steps + 1):]
agg = collections.Counter()
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    end_prev = int(seg[0]["End_Timestamp"])
    tot = 0
    big = []
    for prev, cur in zip(seg[:-1], seg[1:]):
        end_prev = max(end_prev, int(prev["End_Timestamp"]))
        gap = int(cur["Start_Timestamp"]) - end_prev
        if gap > 0:
            tot += gap
        if gap > 40000:
            big.append((gap / 1e3, prev["Kernel_Name"][:60], cur["Kernel_Name"][:60]))
    print("step: idle %.2f ms in total, %d gaps > 40 us:" % (tot / 1e6, len(big)))
    for g, p, c in sorted(big, reverse=True)[:14]:
        print("   %8.1f us   after %-60s before %s" % (g, p, c))
