#!/bin/bash
# round 3, job c: the reworked parity tests (well-posed shapes, diagnostics printed), the device-side
# NMS + memberships-before-download path, full GPU suite, cfg5 line + breakdown
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_fitting_batch_gpu.py tests/test_parity_fullsize_bwd_gpu.py -m gpu -q -s --durations=8 > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
timeout 1500 python -m pytest tests -m gpu -q --durations=5 --deselect tests/test_parity_fullsize_bwd_gpu.py --deselect tests/test_fused_gpu.py --deselect tests/test_fitting_batch_gpu.py > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 1200 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o b -- python3 $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-dense --profile-steps 0 > $O/prof5.log 2>&1
cd $R
python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
timeout 600 python tools/torch_sites.py > $O/torch_sites.txt 2>&1
grep -i "parity\|eval-mode\|passed\|failed\|rc \|Error\|assert" $O/pytest_new.log | head -40; tail -4 $O/pytest.log; cut -c1-400 $O/bench_cfg5.json; tail -3 $O/bench_cfg5.err; cat $O/breakdown.txt; grep -v "amdgpu.ids" $O/torch_sites.txt | head -24
