cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5j
mkdir -p $O
python -m pytest tests/test_workloads_gpu.py -x -q -s -m gpu -k full_batch > $O/pytest.txt 2>&1
grep -v "^$" $O/pytest.txt | tail -40
