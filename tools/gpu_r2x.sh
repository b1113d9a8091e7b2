#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2x
mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/tools/kbench.py knn64 > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -o p -- python3 $GRAFT_REPO_ROOT/tools/kbench.py knn64 > $O/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc3 -o p -- python3 $GRAFT_REPO_ROOT/tools/kbench.py knn64 > $O/pmc3.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
python - <<'P'
import csv, glob, collections
for d in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob("gpurun_out/r2x/%s/*counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "x3_pass1" in n or "pn_knn_mfma_kernel<32, 2, 0, 0>" in n or "pn_knn_mfma_kernel<64, 1, 2, 0>" in n:
                acc[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for n, cs in acc.items():
            print(d, n, {c: "%.3g" % (sum(v) / len(v)) for c, v in cs.items()})
P
