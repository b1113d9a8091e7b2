"""Condense gpurun_out/evidence_r06 (tools/evidence_round6.sh) into the committed summaries of
profiles/ (round 6): python tools/condense_r06.py [gpurun_out/evidence_r06]"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "evidence_r06")


def find(sub, name):
    hits = glob.glob(os.path.join(SRC, sub, "**", name), recursive=True)
    return hits[0] if hits else None


def stats(src, dst, command, note, keep=60):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    n = sum(int(r["Calls"]) for r in rows)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalMs", "AvgUs", "MinUs", "MaxUs", "Percentage"])
        for r in rows[:keep]:
            c, t = int(r["Calls"]), float(r["TotalDurationNs"])
            w.writerow([r["Name"][:140], c, round(t / 1e6, 3), round(t / c / 1e3, 2), round(float(r["MinNs"]) / 1e3, 2),
                        round(float(r["MaxNs"]) / 1e3, 2), round(100 * t / tot, 2)])
        f.write("# %s: %d kernel launches, %.1f ms of kernel time in the whole trace (%s); the %d heaviest kernels kept\n"
                % (command, n, tot / 1e6, note, keep))


def pmc(dirs, dst, names):
    agg = collections.OrderedDict()
    for d in dirs:
        fn = find(d, "p_counter_collection.csv")
        if not fn:
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


def copy(name, dst, head=None, drop=("amdgpu.ids", "UserWarning", "_warn_once", "ROCTracer")):
    src = os.path.join(SRC, name)
    if not os.path.exists(src):
        print("missing", src)
        return
    lines = [ln for ln in open(src, errors="replace").read().splitlines() if not any(d in ln for d in drop)]
    if head:
        lines = lines[:head]
    with open(os.path.join(P, dst), "w") as f:
        f.write("\n".join(lines) + "\n")


PO = "PARSENET_MS_SPARSE=%s rocprofv3 --kernel-trace --stats -- python3 bench.py --profile-only"
for tag, name, cmd, note in (
        ("s5", "cfg5_profile_only", PO % "1", "the pre-trained network from the cache, then ONLY the profiled pass over the pool, planned launches; the DEFAULT: mean-shift backward through the centre rows only"),
        ("s5b", "cfg5_profile_only_dense_backward", "PARSENET_MS_ROWS_BWD=0 " + PO % "1", "as above with the dense backward passes over all rows (callers with a dense gradient), planned launches"),
        ("s5d", "cfg5_profile_only_dense", "PARSENET_MS_ROWS_BWD=0 " + PO % "0", "dense backward passes AND dense mean-shift launches"),
        ("s4", "cfg4", "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0", "pool pass, warm-up and timed steps"),
        ("s_cfg2", "cfg2", "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0", "warm-up and timed steps"),
        ("s_cfg3", "cfg3", "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0", "warm-up and timed steps")):
    f = find(tag, "b_kernel_stats.csv")
    if f:
        stats(f, os.path.join(P, "r06_%s_kernel_stats.csv" % name), cmd, note)
for j in ("bench_cfg5", "bench_cfg5_b", "bench_cfg5_c", "bench_cfg4", "bench_cfg2", "bench_cfg3", "po_stats", "po_stats_bwd", "po_stats_dense"):
    src = os.path.join(SRC, j + ".json")
    if os.path.exists(src):
        lines = [ln for ln in open(src).read().splitlines() if ln.startswith("{")]
        if lines:
            out = {"po_stats": "bench_cfg5_profile_only_under_rocprofv3",
                   "po_stats_bwd": "bench_cfg5_profile_only_dense_backward_under_rocprofv3",
                   "po_stats_dense": "bench_cfg5_profile_only_dense_under_rocprofv3"}.get(j, j)
            open(os.path.join(P, "r06_" + out + ".json"), "w").write(lines[-1] + "\n")
pmc(["pmc_FETCH_SIZE_rows", "pmc_WRITE_SIZE_rows", "pmc_SQ_rows", "pmc_GRBM_GUI_ACTIVE_rows"], os.path.join(P, "r06_meanshift_x3_planned_fwd_only_cfg5_pmc.csv"), ("pn_ms3_kernel",))
pmc(["pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ"], os.path.join(P, "r06_meanshift_x3_planned_cfg5_pmc.csv"), ("pn_ms3_kernel",))
pmc(["pmc_FETCH_SIZE_dense", "pmc_WRITE_SIZE_dense", "pmc_SQ_dense"], os.path.join(P, "r06_meanshift_x3_dense_cfg5_pmc.csv"), ("pn_ms3_kernel",))
copy("roofline_check.txt", "r06_roofline_check.txt")
copy("roofline_check_bwd.txt", "r06_roofline_check_dense_backward.txt")
copy("roofline_check_dense.txt", "r06_roofline_check_dense.txt")
copy("timeline.txt", "r06_cfg5_step_timeline.txt")
copy("breakdown.txt", "r06_cfg5_step_breakdown.txt")
copy("gaps.txt", "r06_cfg5_step_gaps.txt", head=60)
pmc(["pmc_named_TCC_HIT_sum", "pmc_named_FETCH_SIZE", "pmc_named_WRITE_SIZE", "pmc_named_SQ_WAVE_CYCLES"],
    os.path.join(P, "r06_named_kernels_pmc.csv"), ("pn_edgeconv_reduce_kernel", "pn_edgeconv_bwd_gather_kernel", "pn_knn_smallk_kernel", "pn_rev_"))
copy("torch_sites.txt", "r06_cfg5_torch_sites.txt", head=110)
copy("host_cprofile.txt", "r06_cfg5_host_cprofile.txt", head=70)
copy("op_lines.txt", "r06_op_lines.txt", head=110)
copy("copy_census.txt", "r06_copy_census.txt", head=110)
copy("host.txt", "r06_host.txt")
copy("kbench.log", "r06_named_kernels_kbench.txt")
copy("clock_sysfs.txt", "r06_clock_sysfs_sampler.txt")
src = os.path.join(SRC, "bench_cfg5_forced_collective.json")
if os.path.exists(src):
    lines = [ln for ln in open(src).read().splitlines() if ln.startswith("{")]
    if lines:
        open(os.path.join(P, "r06_bench_cfg5_forced_collective.json"), "w").write(lines[-1] + "\n")
det = []
for w in ("cfg2", "cfg3", "cfg4", "cfg5"):
    f = os.path.join(SRC, "det_%s.txt" % w)
    if os.path.exists(f):
        det.append("# python tools/determinism_probe.py --workload %s%s --steps 3" % (w, " --pretrain 40" if w == "cfg5" else ""))
        det += [ln for ln in open(f, errors="replace").read().splitlines() if "amdgpu.ids" not in ln]
if det:
    open(os.path.join(P, "r06_determinism_probe.txt"), "w").write("\n".join(det) + "\n")
log = os.path.join(SRC, "pytest.log")
if os.path.exists(log):
    txt = open(log, errors="replace").read()
    keep = [ln[:1600] for ln in txt.splitlines()
            if re.search(r"parity:|eval-mode control points|passed|failed|^rc |s call |16 groups|pinned-graph whole step|"
                         r"relative error of every layer|edge conv \d|conv\d \+ bn|control grid|against float64|"
                         r"smallest gradient cosine|rows of the product", ln)]
    open(os.path.join(P, "r06_gpu_suite.txt"), "w").write(
        "# python -m pytest tests -m gpu -q -s --durations=10 on the evidence box (tools/evidence_round6.sh)\n" +
        "\n".join(keep) + "\n")
# reproducibility of the headline: what two processes of the driver's command print
rep = []
for j in ("bench_cfg5", "bench_cfg5_b", "bench_cfg5_c"):
    src = os.path.join(SRC, j + ".json")
    if os.path.exists(src):
        lines = [ln for ln in open(src).read().splitlines() if ln.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            r = d.get("roofline") or {}
            rep.append("%s: value %.2f shapes/s (%.2f ms per step), value_dense %s, pretrain_final_loss %r, clusters_per_shape %s, "
                       "tile pairs executed (forward pass) %s, frac %s" % (
                           j, d["value"], d["ms_per_step"], d.get("value_dense"), d["config"].get("pretrain_final_loss"),
                           d["config"].get("clusters_per_shape"),
                           (r.get("passes") or {}).get("meanshift_fwd", {}).get("tile_pairs_executed"), r.get("frac")))
if rep:
    open(os.path.join(P, "r06_bench_reproducibility.txt"), "w").write(
        "# three processes of `python bench.py` on one box: the pre-training, the clustering and the executed work are the\n"
        "# same bit for bit (the value moves with the host: the boxes are shared)\n" + "\n".join(rep) + "\n")
print(sorted(f for f in os.listdir(P) if f.startswith("r06_")))
