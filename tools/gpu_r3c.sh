#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $O
for G in 512 1024; do
export PN_MS_FLAT_G=$G
MS_PROBE_EMB=$GRAFT_REPO_ROOT/tools/dbg/ms_emb.pt timeout 300 python3 $GRAFT_REPO_ROOT/tools/ms_probe.py 2>&1 | grep -A3 "sparse=1"
for C in FETCH_SIZE WRITE_SIZE; do
MS_PROBE_EMB=$GRAFT_REPO_ROOT/tools/dbg/ms_emb.pt timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- python3 $GRAFT_REPO_ROOT/tools/ms_probe.py > $O/pmc_$C.log 2>&1
find $O/pmc_$C -name "*kernel_trace.csv" -delete
done
cd $GRAFT_REPO_ROOT
python tools/condense_r02.py pmc ../gpurun_out/r3c/ms_pmc_$G.csv pn_ms3_kernel $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
echo "G=$G"; cat gpurun_out/r3c/ms_pmc_$G.csv | grep -v "245760\|323584"
cd /tmp
done
