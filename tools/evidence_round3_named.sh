#!/bin/bash
# round 3: the north_star's named kernels again (edge-feature gather, fused edge-conv gather-reduce,
# Chamfer NN) after the neighbour lists went into registers: kbench timings, rocprofv3 kernel
# statistics and PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit rate / wait share) of tools/evidence_kernels.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3named
mkdir -p $O
timeout 600 python -m pytest tests/test_edgeconv_gpu.py tests/test_golden_gpu.py tests/test_chamfer_gpu.py tests/test_encoder_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 300 python tools/kbench.py edge chamfer > $O/kbench.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o k -- python3 $R/tools/evidence_kernels.py > $O/kstats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc1 -o p -- python3 $R/tools/evidence_kernels.py > $O/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc2 -o p -- python3 $R/tools/evidence_kernels.py > $O/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc3 -o p -- python3 $R/tools/evidence_kernels.py > $O/pmc3.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc4 -o p -- python3 $R/tools/evidence_kernels.py > $O/pmc4.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete
tail -2 $O/pytest.log; grep -v "amdgpu.ids\|Warn" $O/kbench.log | tail -16; grep "pn_edge_feature\|pn_edgeconv_reduce\|pn_chamfer" $O/kstats/k_kernel_stats.csv | cut -c1-60,200-330
