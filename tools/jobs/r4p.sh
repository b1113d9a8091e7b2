cd $GRAFT_REPO_ROOT
SEQ="tests/test_fitting_batch_gpu.py tests/test_e2e_gpu.py tests/test_fullsize_gpu.py tests/test_golden_gpu.py tests/test_parity_fullsize_bwd_gpu.py"
for rep in 1 2; do
echo "== rep $rep"
timeout 1500 python -m pytest $SEQ -x -q -s 2>&1 | grep -v amdgpu | grep "loop parity: res\|passed\|failed\|Error\|differs" | cut -c1-400
done
