"""Per-step GPU time of the LAST steps of a rocprofv3 kernel trace of bench.py, grouped by stage
(kernel-name patterns), and the GPU idle time inside a step.  python tools/step_breakdown.py trace.csv [steps] [detail]
("detail": under every group its kernels by name — launches and ms per step, grid size of the heaviest launch)"""
import csv
import sys
import collections

fn = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the Adam kernel (multi_tensor_apply) bursts: find the last `steps` optimizer bursts
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "multi_tensor_apply" in n or "adam" in n.lower()]
bursts = []
for i in adam:
    if not bursts or i - bursts[-1][-1] > 50:
        bursts.append([i])
    else:
        bursts[-1].append(i)
ends = [b[-1] for b in bursts][-(steps + 1):]
groups = collections.OrderedDict([
    ("mean-shift iterations", ("pn_ms3_", "pn_ms_", "pn_msh_")),
    ("bandwidth / nms selection", ("pn_dotsel", "pn_sel", "dot_select", "pn_dot")),
    ("kNN", ("pn_knn",)),
    ("edge conv / group norm (HIP)", ("pn_edge", "pn_ecb", "pn_gn", "pn_transpose", "pn_moments", "pn_rev_")),
    ("fused glue: triplet / memberships / affine / Adam / gather (HIP)", ("pn_triplet", "pn_member", "pn_affine", "pn_adam", "pn_gather_flat", "pn_wmax", "pn_nms", "pn_kmeans", "pn_stats")),
    ("batched fits (HIP)", ("pn_wmom", "pn_primfit", "pn_cone", "pn_prim_residual", "pn_bspline", "pn_chamfer")),
    ("GEMM (rocBLAS / hipBLASLt; pn_gemm_x3: bf16 x 3)", ("Cijk", "gemm", "rocblas", "pn_gx", "gx_")),
    ("other hand-written kernels (pn_*)", ("pn_",)),
    ("elementwise / reductions / sort (torch)", ("",)),
])
detail = len(sys.argv) > 3 and sys.argv[3] == "detail"
tot = collections.Counter()
cnt = collections.Counter()
per = collections.defaultdict(lambda: [0, 0, 0, ""])      # (group, name) -> ns, launches, longest, its grid
busy = 0
span = 0
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    span += int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    for r in seg:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy += d
        for g, pats in groups.items():
            if any(p in r["Kernel_Name"] for p in pats):
                tot[g] += d
                cnt[g] += 1
                e = per[(g, r["Kernel_Name"][:110])]
                e[0] += d
                e[1] += 1
                if d > e[2]:
                    e[2] = d
                    e[3] = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
                break
n = len(ends) - 1
print("steps analysed: %d; wall per step (first to last kernel) %.2f ms; kernel time %.2f ms; launches per step %.0f"
      % (n, span / n / 1e6, busy / n / 1e6, sum(cnt.values()) / n))
for g in groups:
    print("  %-42s %7.2f ms  %6.0f launches" % (g, tot[g] / n / 1e6, cnt[g] / n))
    if detail:
        mine = sorted(((k[1], v) for k, v in per.items() if k[0] == g), key=lambda t: -t[1][0])
        for name, (ns, c, longest, grid) in mine[:14]:
            print("      %7.3f ms %6.1f x  longest %7.1f us (grid %s)  %s" % (ns / n / 1e6, c / n, longest / 1e3, grid, name))
