# cfg5: where does the wall time of a step go now?  kernel trace of timed steps -> timeline / breakdown / gaps / torch sites
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5g
rm -rf $O; mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r05.pt
python bench.py --no-cpu-baseline --no-dense > $O/b0.json 2> $O/b0.err
python -c "import json;d=json.load(open('$O/b0.json'));print('untraced', round(d['value'],2), round(d['ms_per_step'],2))"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o b -- python3 $R/bench.py --no-cpu-baseline --no-dense --profile-steps 0 --steps 8 --warmup 3 > $O/bench_traced.json 2> $O/bench_traced.err
cd $R
T=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
python tools/step_timeline.py $T 1 1 > $O/timeline.txt 2>&1
python tools/step_breakdown.py $T 5 > $O/breakdown.txt 2>&1
python tools/step_gaps.py $T 2 > $O/gaps.txt 2>&1
rm -rf $O/tr
cat $O/breakdown.txt; head -60 $O/timeline.txt
python tools/host_cprofile.py > $O/host_cprofile.txt 2>&1
head -45 $O/host_cprofile.txt
