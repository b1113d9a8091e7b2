# small-k kernel: phase timers (wave 0 of every workgroup) on the shapes of the SplineNet layers
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5n
mkdir -p $O
PN_EXTRA_HIPCC_FLAGS="-DKSK_TIMERS" python -m parsenet_codebase_amd.build > $O/build.txt 2>&1
for S in "6 256 5000" "5 256 5000" "6 128 5000" "6 64 5000" "12 64 5000" "6 3 5000"; do
  python tools/probes/ksk_timers.py $S 2>&1 | grep -v amdgpu.ids | tee -a $O/timers.txt
done
# per-kernel detail of a traced cfg5 step (default build)
python -m parsenet_codebase_amd.build > $O/build2.txt 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-dense --profile-steps 0 --steps 8 --warmup 3 > $O/bench_traced.json 2> $O/bench_traced.err
cd $GRAFT_REPO_ROOT
T=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
python tools/step_breakdown.py $T 5 detail > $O/breakdown_detail.txt 2>&1
rm -rf $O/tr
# cfg2 / cfg3 / cfg4: gradients gathered into the bucket (default) against added into its views, alternating
for i in 1 2 3; do for G in 1 0; do for W in cfg2 cfg3; do
  PARSENET_BUCKET_GATHER=$G timeout 300 python bench.py --workload $W --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W gather=$G', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a $O/bucket_ab.txt
done; done; done
