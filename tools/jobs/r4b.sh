set -x
python tools/determinism_probe.py --workload cfg3 --steps 3 > gpurun_out/det_cfg3.txt 2>&1
python tools/determinism_probe.py --workload cfg2 --steps 3 > gpurun_out/det_cfg2.txt 2>&1
python tools/determinism_probe.py --workload cfg4 --steps 3 > gpurun_out/det_cfg4.txt 2>&1
python tools/determinism_probe.py --workload cfg5 --pretrain 40 --steps 3 > gpurun_out/det_cfg5.txt 2>&1
for f in gpurun_out/det_cfg*.txt; do tail -n 6 $f; done
timeout 1200 python -m pytest tests/test_determinism_gpu.py tests/test_edgeconv_gpu.py tests/test_fused_gpu.py tests/test_fitting_batch_gpu.py tests/test_chamfer_gpu.py -x -q > gpurun_out/gpu_suite_b.txt 2>&1; tail -n 25 gpurun_out/gpu_suite_b.txt
