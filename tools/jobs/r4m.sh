# host side of a cfg5 step: the host step of standardize_segments on this box, and where the GPU idles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4m
mkdir -p $O
cat > /tmp/rot.py <<'P'
import time, torch, numpy as np, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from parsenet_codebase_amd.fitting_batch import host_minor_axis_rotations
g = torch.Generator().manual_seed(0)
A = torch.randn(8, 50, 3, generator=g); cov = A.transpose(1, 2) @ A
Fm = np.random.rand(3, 3)
for rep in range(3):
    t = time.perf_counter()
    for _ in range(20): host_minor_axis_rotations(cov)
    a = (time.perf_counter() - t) / 20 * 1e3
    t = time.perf_counter()
    for _ in range(200): np.linalg.inv(Fm)
    b = (time.perf_counter() - t) / 200 * 1e6
    t = time.perf_counter()
    for _ in range(50): torch.linalg.eig(cov)
    c = (time.perf_counter() - t) / 50 * 1e6
    print("rotations of 8: %.3f ms; inv %.1f us; eig(8) %.1f us; torch threads %d" % (a, b, c, torch.get_num_threads()))
P
(echo default; python /tmp/rot.py; echo OPENBLAS_NUM_THREADS=1; OPENBLAS_NUM_THREADS=1 python /tmp/rot.py; echo OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1; OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 python /tmp/rot.py) > $O/rot.txt 2>&1
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r04.pt
python bench.py --no-cpu-baseline --no-dense > $O/bench_plain.json 2> $O/bench_plain.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o b -- python3 $R/bench.py --no-cpu-baseline --no-dense --profile-steps 0 --steps 8 --warmup 3 > $O/bench_traced.json 2> $O/bench_traced.err
cd $R
T=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
python tools/step_timeline.py $T 2 1 > $O/timeline.txt 2>&1
python tools/step_gaps.py $T 3 > $O/gaps.txt 2>&1
python tools/step_breakdown.py $T 5 > $O/breakdown.txt 2>&1
rm -rf $O/tr
cat $O/rot.txt; cut -c1-200 $O/bench_plain.json; cut -c1-200 $O/bench_traced.json; cat $O/breakdown.txt; head -70 $O/timeline.txt
