cd $GRAFT_REPO_ROOT
SEQ="tests/test_fitting_batch_gpu.py tests/test_e2e_gpu.py tests/test_fullsize_gpu.py tests/test_golden_gpu.py tests/test_parity_fullsize_bwd_gpu.py"
for mode in 1 0; do
echo "== PARSENET_MS_NEAREST=$mode"
PARSENET_MS_NEAREST=$mode timeout 1200 python -m pytest $SEQ -x -q -s -k "not whole_e2e_step and not meanshift_backward_at and not splinenet_full" 2>&1 | grep -v amdgpu | grep "loop parity\|passed\|failed\|Error\|differs" | cut -c1-400
done
