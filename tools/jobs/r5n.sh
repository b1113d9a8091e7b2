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
