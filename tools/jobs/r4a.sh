set -x
python tools/determinism_probe.py --workload cfg4 --steps 3 --warn > gpurun_out/det_cfg4.txt 2>&1
python tools/determinism_probe.py --workload cfg2 --steps 3 --warn > gpurun_out/det_cfg2.txt 2>&1
python tools/determinism_probe.py --workload cfg3 --steps 3 > gpurun_out/det_cfg3.txt 2>&1
python tools/determinism_probe.py --workload cfg5 --pretrain 40 --steps 3 --warn > gpurun_out/det_cfg5.txt 2>&1
tail -5 gpurun_out/det_cfg*.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.txt 2>&1; tail -15 gpurun_out/gpu_suite.txt
