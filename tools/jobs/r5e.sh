cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5e
mkdir -p $O
python -m pytest tests/test_edgeconv_gpu.py tests/test_encoder_gpu.py tests/test_determinism_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
python tools/kbench.py edge > $O/kbench_edge.txt 2>&1
grep edgeconv $O/kbench_edge.txt
