# after the full evidence run: HBM traffic of the default process's planned launches (forward-only plans) in passes of their
# own, then the three cfg5 lines again (bench.py now quotes `traffic` from that file) — written where condense_r05.py looks
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence_r05
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r05.pt
timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1      # (pre-trains once and fills the cache)
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
PARSENET_MS_SPARSE=1 timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${C}_rows -o p -- python3 $R/bench.py --profile-only > /dev/null 2> $O/po_${C}_rows.err
done
PARSENET_MS_SPARSE=1 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_SQ_rows -o p -- python3 $R/bench.py --profile-only > $O/po_sq_rows.json 2> $O/po_sq_rows.err
find $O -name "*kernel_trace.csv" -delete
cd $R
python tools/condense_r05.py > /dev/null 2>&1
unset PARSENET_PRETRAIN_CACHE
timeout 1200 python bench.py > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 1200 python bench.py --no-cpu-baseline > $O/bench_cfg5_b.json 2> $O/bench_cfg5_b.err
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r05.pt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cfg5_c.json 2> $O/bench_cfg5_c.err
for f in bench_cfg5 bench_cfg5_b bench_cfg5_c; do python - <<PY
import json
d=[json.loads(l) for l in open("$O/$f.json") if l.startswith("{")][-1]
print("$f", round(d["value"],1), round(d["ms_per_step"],2), d["roofline"]["frac"], d["roofline"]["traffic"])
PY
done
# cfg2 / cfg3 again (host-bound steps: 4.3 ... 5.4 ms over the day's boxes), three times each
for W in cfg2 cfg3; do for i in 1 2 3; do
timeout 300 python bench.py --workload $W --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a $O/cfg23_again.txt
done; done
