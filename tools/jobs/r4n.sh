# is the host side throttled?  cgroup CPU quota / throttling counters around bench runs with torch's default
# intra-op thread count (one per visible core) and with one thread
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4n
mkdir -p $O
stat() { echo "--- $1"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "usage_usec|nr_periods|nr_throttled|throttled_usec"; }
(echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; nproc; grep Cpus_allowed_list /proc/self/status; cat /proc/loadavg) > $O/host.txt 2>&1
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r04.pt
python bench.py --no-cpu-baseline --no-dense > /dev/null 2>&1      # creates the cache
{
stat start
for V in 0 1 0 1; do
  python bench.py --no-cpu-baseline --no-dense --host-threads $V > $O/b_$V.json 2> $O/b_$V.err
  echo "host-threads $V: $(python -c "import json;d=json.load(open('$O/b_$V.json'));print(round(d['value'],2), round(d['ms_per_step'],2))")"
  stat "after host-threads $V"
done
OMP_NUM_THREADS=1 python bench.py --no-cpu-baseline --no-dense > $O/b_omp1.json 2> $O/b_omp1.err
echo "OMP_NUM_THREADS=1: $(python -c "import json;d=json.load(open('$O/b_omp1.json'));print(round(d['value'],2), round(d['ms_per_step'],2))")"
stat "after OMP_NUM_THREADS=1"
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 python bench.py --no-cpu-baseline --no-dense > $O/b_all1.json 2> $O/b_all1.err
echo "OMP/OPENBLAS/MKL=1: $(python -c "import json;d=json.load(open('$O/b_all1.json'));print(round(d['value'],2), round(d['ms_per_step'],2))")"
stat "after all 1"
cat /proc/loadavg
} > $O/ab.txt 2>&1
cat $O/host.txt $O/ab.txt
