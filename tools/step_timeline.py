"""Where inside a cfg5 step the GPU idles: the last steps of a rocprofv3 kernel trace of bench.py cut
into bins of 2 ms; per bin the busy share and the kernel family that owns most of it.
python tools/step_timeline.py trace.csv [steps] [bin_ms]"""
import csv
import sys
import collections

fn = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
binw = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 2e6
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
fam = [("ms3", ("pn_ms3_", "pn_ms_")), ("knn/sel", ("pn_knn", "pn_dot", "pn_sel")),
       ("edge/gn", ("pn_edge", "pn_ecb", "pn_gn", "pn_transpose", "pn_rev_")),
       ("fit", ("pn_wmom", "pn_primfit", "pn_cone", "pn_prim_residual", "pn_bspline", "pn_chamfer", "pn_member",
                "pn_nms", "pn_triplet", "pn_affine", "pn_gather", "pn_fit")),
       ("gemm", ("Cijk", "gemm", "rocblas")), ("torch", ("",))]


def family(n):
    for f, pats in fam:
        if any(p in n for p in pats):
            return f


for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in seg)
    nb = int((t1 - t0) / binw) + 1
    busy = [collections.Counter() for _ in range(nb)]
    launches = [0] * nb
    for r in seg:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        f = family(r["Kernel_Name"])
        launches[int(s / binw)] += 1
        i = int(s / binw)
        while s < e:
            hi = min(e, (i + 1) * binw)
            busy[i][f] += hi - s
            s = hi
            i += 1
    print("step of %.2f ms (first kernel to last): bin start ms, busy %%, launches, families by time" % ((t1 - t0) / 1e6))
    for i in range(nb):
        tot = sum(busy[i].values())
        print("  %6.1f  %5.1f %%  %4d   %s" % (i * binw / 1e6, 100.0 * tot / binw, launches[i],
              "  ".join("%s %.2f" % (f, v / 1e6) for f, v in busy[i].most_common(3))))
