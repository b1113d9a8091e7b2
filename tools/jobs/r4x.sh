# host threads for the small steps (cfg2) and the large one (cfg5); then the full GPU suite with the oracle's threads intact
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4x
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r04.pt
python bench.py --no-cpu-baseline --no-dense > /dev/null 2>&1
for rep in 1 2; do
for T in 1 2 4 8 0; do
  for W in cfg2 cfg5; do
    python bench.py --workload $W --no-cpu-baseline --no-dense --host-threads $T > $O/b.json 2> $O/b.err
    echo "host-threads $T $W: $(python -c "import json;d=json.load(open('$O/b.json'));print(round(d['value'],2), round(d['ms_per_step'],3))")"
  done
done
done > $O/threads.txt 2>&1
cat $O/threads.txt
timeout 1500 python -m pytest tests -m gpu -q -s --durations=10 > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
