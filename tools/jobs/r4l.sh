cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence_r04
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r04.pt
python bench.py --no-cpu-baseline --no-dense > /dev/null 2>&1    # (fresh box: re-create the cache)
rm -rf $O/pmc_SQ_dense $O/s5d $O/pmc_FETCH_dense $O/pmc_WRITE_dense
cd /tmp
PARSENET_MS_SPARSE=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_SQ_dense -o p -- python3 $R/bench.py --profile-only > $O/po_sq_dense.json 2> $O/po_sq_dense.err
PARSENET_MS_SPARSE=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5d -o b -- python3 $R/bench.py --profile-only > $O/po_stats_dense.json 2> $O/po_stats_dense.err
for C in FETCH_SIZE WRITE_SIZE; do
PARSENET_MS_SPARSE=0 timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${C}_dense -o p -- python3 $R/bench.py --profile-only > /dev/null 2>&1
done
cd $R
find $O -name "*kernel_trace.csv" -delete
Kd=$(find $O/s5d -name "b_kernel_stats.csv" | head -1); Cd=$(find $O/pmc_SQ_dense -name "p_counter_collection.csv" | head -1)
python tools/roofline_check.py $O/po_stats_dense.json $Kd $Cd $O/roofline_check_dense.txt
