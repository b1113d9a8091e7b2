#!/bin/bash
mkdir -p gpurun_out/r2t
timeout 900 python -m pytest tests/test_meanshift_gpu.py -m gpu -x -q > gpurun_out/r2t/pytest_ms.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r2t/pytest_ms.log
MS_PROBE_EMB=tools/dbg/ms_emb.pt timeout 600 python tools/ms_probe.py > gpurun_out/r2t/probe.log 2>&1
echo "probe rc $?"
tail -40 gpurun_out/r2t/probe.log
