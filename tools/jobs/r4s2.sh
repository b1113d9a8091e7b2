# the two SplineNets of a fitting stage on two streams (PARSENET_SPLINE_STREAMS=1) against one after the other (0)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4s2
mkdir -p $O
python -m pytest tests/test_fitting_batch_gpu.py tests/test_determinism_gpu.py tests/test_e2e_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r04.pt
python bench.py --no-cpu-baseline --no-dense > /dev/null 2>&1
for V in 1 0 1 0 1 0; do
  PARSENET_SPLINE_STREAMS=$V python bench.py --no-cpu-baseline --no-dense > $O/b_$V.json 2> $O/b_$V.err
  echo "PARSENET_SPLINE_STREAMS=$V: $(python -c "import json;d=json.load(open('$O/b_$V.json'));print(round(d['value'],2), round(d['ms_per_step'],2), 'clusters', d['config']['clusters_per_shape'], 'segments', d['config']['segments_per_shape'])")"
done
python tools/determinism_probe.py --workload cfg5 --pretrain 40 --steps 3 2>&1 | tail -1
