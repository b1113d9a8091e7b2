cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5k
mkdir -p $O
for S in 13 51 55 56 58 89; do
PARITY_SPLINE_SHAPE=$S python -m pytest tests/test_parity_fullsize_bwd_gpu.py -x -q -s -m gpu -k pinned_graphs > $O/pytest_$S.txt 2>&1
grep "^pinned-graph whole step" $O/pytest_$S.txt | cut -c1-900
done
