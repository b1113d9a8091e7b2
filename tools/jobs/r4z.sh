# forward pass of the mean-shift iterations: ping-pong schedule (PN_MS_PINGPONG=2) against the default, planned launches, same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4z
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r04.pt
python bench.py --no-cpu-baseline --no-dense > /dev/null 2>&1
for V in 2 1 2 1; do
  PN_MS_PINGPONG=$V python bench.py --no-cpu-baseline --no-dense > $O/b_$V.json 2> $O/b_$V.err
  echo "PN_MS_PINGPONG=$V: $(python -c "import json;d=json.load(open('$O/b_$V.json'));p=d['roofline']['passes']['meanshift_fwd'];print(round(d['value'],2), round(d['ms_per_step'],2), 'forward pass ms per launch', round(p['avg_launch_ms'],4), 'frac', round(p['frac'],4), 'rows_bwd', d['kernel_ms_per_step']['meanshift_rows_bwd'])")"
done
