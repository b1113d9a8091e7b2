"""Condense gpurun_out/evidence_r03 (tools/evidence_round3.sh) into the committed summaries of
profiles/ (round 3): python tools/condense_r03.py [gpurun_out/evidence_r03]"""
import csv
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "evidence_r03")


def stats(src, dst, command, keep=60):
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
        f.write("# %s: %d kernel launches, %.1f ms of kernel time in the whole trace (pre-pass over the pool, warm-up "
                "and timed steps; per-step figures: r03_cfg5_step_breakdown.txt); the %d heaviest kernels kept\n"
                % (command, n, tot / 1e6, keep))


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


for tag, cmd in (("s5", "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload cfg5 --steps 10 --warmup 3 "
                        "--no-cpu-baseline --no-dense --profile-steps 0"),
                 ("s4", "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload cfg4 --steps 10 --warmup 3 "
                        "--no-cpu-baseline --profile-steps 0"),
                 ("s_cfg2", "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload cfg2 --steps 20 --warmup 5 "
                            "--no-cpu-baseline --profile-steps 0"),
                 ("s_cfg3", "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload cfg3 --steps 20 --warmup 5 "
                            "--no-cpu-baseline --profile-steps 0")):
    f = os.path.join(SRC, tag, "b_kernel_stats.csv")
    if os.path.exists(f):
        name = {"s5": "cfg5", "s4": "cfg4", "s_cfg2": "cfg2", "s_cfg3": "cfg3"}[tag]
        stats(f, os.path.join(P, "r03_%s_kernel_stats.csv" % name), cmd)
for j in ("bench_cfg5", "bench_cfg5_b", "bench_cfg5_planned", "bench_cfg4", "bench_cfg2", "bench_cfg3"):
    src = os.path.join(SRC, j + ".json")
    if os.path.exists(src):
        lines = [ln for ln in open(src).read().splitlines() if ln.startswith("{")]
        if lines:
            open(os.path.join(P, "r03_" + j + ".json"), "w").write(lines[-1] + "\n")
copy("breakdown.txt", "r03_cfg5_step_breakdown.txt")
copy("torch_sites.txt", "r03_cfg5_torch_sites.txt", head=48)
copy("host_cprofile.txt", "r03_cfg5_host_cprofile.txt", head=70)
copy("host.txt", "r03_host.txt")
# the GPU suite: summary line, slowest tests, and the diagnostics the full-size parity tests print
log = os.path.join(SRC, "pytest.log")
if os.path.exists(log):
    txt = open(log, errors="replace").read()
    keep = [ln[:1200] for ln in txt.splitlines()
            if re.search(r"parity:|eval-mode control points|passed|failed|^rc |s call ", ln)]
    open(os.path.join(P, "r03_gpu_suite.txt"), "w").write(
        "# python -m pytest tests -m gpu -q -s --durations=10 on the evidence box (tools/evidence_round3.sh)\n" +
        "\n".join(keep) + "\n")
pmc = os.path.join(SRC, "r03_meanshift_x3_planned_cfg5_pmc.csv")
if os.path.exists(pmc):
    shutil.copy(pmc, os.path.join(P, "r03_meanshift_x3_planned_cfg5_pmc.csv"))
print(sorted(f for f in os.listdir(P) if f.startswith("r03_")))
