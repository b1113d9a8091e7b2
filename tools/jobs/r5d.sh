# int32 graph + XCD-aware gather mapping: parity tests of the graph / edge-conv family, named-kernel bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5d
mkdir -p $O
python -m pytest tests/test_knn_gpu.py tests/test_edgeconv_gpu.py tests/test_encoder_gpu.py tests/test_fused_gpu.py tests/test_determinism_gpu.py tests/test_golden_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
python tools/kbench.py edge > $O/kbench_edge.txt 2>&1
grep edgeconv $O/kbench_edge.txt
python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest2.txt 2>&1
tail -5 $O/pytest2.txt
