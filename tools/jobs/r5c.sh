# counters of the small-k kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5c
rm -rf $O; mkdir -p $O
for SHAPE in "6 3 5000 10" "6 64 5000 10" "6 256 5000 10"; do
  T=$(echo $SHAPE | tr ' ' '_')
  for CS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
    D=$O/${T}_$(echo $CS | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $CS --kernel-trace --output-format csv -d $D -o p -- python3 tools/probes/knn_one.py $SHAPE 3 > /dev/null 2>$D.err
    python3 - "$D" "$T" <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "smallk_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(tag, k, "%.4g per launch (%d launches)" % (v / n, n))
PY
  done
done 2>&1 | tee $O/summary.txt
