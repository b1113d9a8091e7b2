#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2g
mkdir -p $O
timeout 600 python -m pytest tests/test_chamfer_gpu.py tests/test_fitting_batch_gpu.py tests/test_fitting_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 300 python tools/kbench.py chamfer > $O/kbench.log 2>&1
for q in 1 2 4; do echo "Q=$q" >> $O/kbench.log; PN_CHAMFER_Q=$q timeout 300 python tools/kbench.py chamfer >> $O/kbench.log 2>&1; done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kstats -o k -- python3 $GRAFT_REPO_ROOT/tools/evidence_kernels.py > $GRAFT_REPO_ROOT/$O/kstats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/tools/evidence_kernels.py > $GRAFT_REPO_ROOT/$O/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -o p -- python3 $GRAFT_REPO_ROOT/tools/evidence_kernels.py > $GRAFT_REPO_ROOT/$O/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc3 -o p -- python3 $GRAFT_REPO_ROOT/tools/evidence_kernels.py > $GRAFT_REPO_ROOT/$O/pmc3.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc4 -o p -- python3 $GRAFT_REPO_ROOT/tools/evidence_kernels.py > $GRAFT_REPO_ROOT/$O/pmc4.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc5 -o p -- python3 $GRAFT_REPO_ROOT/tools/evidence_kernels.py > $GRAFT_REPO_ROOT/$O/pmc5.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -size +20M -delete
tail -5 $O/pytest.log; cat $O/kbench.log; ls $O/*; tail -2 $O/pmc1.log
