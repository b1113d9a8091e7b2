#!/bin/bash
# round 3, job b: fused kernels + full-size parity tests, the whole GPU suite, the honest cfg5 line
# (held-out pool, 2000 pre-training steps, auto sparse/dense), kernel trace + step breakdown
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
(free -g; nproc; lscpu | grep "Model name") > $O/host.txt 2>&1
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_parity_fullsize_bwd_gpu.py -m gpu -q --durations=12 > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
timeout 1500 python -m pytest tests -m gpu -q --durations=8 --deselect tests/test_parity_fullsize_bwd_gpu.py --deselect tests/test_fused_gpu.py > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 1200 python bench.py --workload cfg5 --steps 20 --warmup 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o b -- python3 $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-dense --profile-steps 0 > $O/prof5.log 2>&1
cd $R
python tools/step_breakdown.py $O/s5/b_kernel_trace.csv > $O/breakdown.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
timeout 600 python tools/torch_sites.py > $O/torch_sites.txt 2>&1
cat $O/host.txt; tail -25 $O/pytest_new.log; tail -6 $O/pytest.log; cut -c1-1500 $O/bench_cfg5.json; tail -3 $O/bench_cfg5.err; cat $O/breakdown.txt; grep -v "amdgpu.ids" $O/torch_sites.txt | head -40
