# cfg5 with the row-restricted mean-shift backward (default) against the dense backward passes, same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4r
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r04.pt
python bench.py --no-cpu-baseline --no-dense > $O/b_rows_0.json 2> $O/b_rows_0.err
for V in 1 0 1 0; do
  PARSENET_MS_ROWS_BWD=$V python bench.py --no-cpu-baseline --no-dense > $O/b_$V.json 2> $O/b_$V.err
  echo "PARSENET_MS_ROWS_BWD=$V: $(python -c "import json;d=json.load(open('$O/b_$V.json'));print(round(d['value'],2), round(d['ms_per_step'],2), d['roofline']['kernel'], round(d['roofline']['frac'],3), {k:v[0] for k,v in list(d['kernel_ms_per_step'].items())[:8]})")"
done > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o b -- python3 $R/bench.py --no-cpu-baseline --no-dense --profile-steps 0 --steps 8 --warmup 3 > $O/bench_traced.json 2> $O/bench_traced.err
cd $R
T=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
python tools/step_timeline.py $T 1 1 > $O/timeline.txt 2>&1
python tools/step_breakdown.py $T 5 > $O/breakdown.txt 2>&1
python tools/step_gaps.py $T 2 > $O/gaps.txt 2>&1
rm -rf $O/tr
cat $O/breakdown.txt; head -50 $O/timeline.txt
