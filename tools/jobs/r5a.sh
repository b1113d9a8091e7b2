# round 5: small-k one-pass kNN kernel — parity + micro-benchmark
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5a
mkdir -p $O
python -m pytest tests/test_knn_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
python tools/kbench.py smallk > $O/kbench.txt 2>&1
cat $O/kbench.txt
