#!/bin/bash
mkdir -p gpurun_out/r2s
timeout 900 python tools/ms_probe.py gpurun_out/r2s/ms_emb.pt > gpurun_out/r2s/probe.log 2>&1
echo "probe rc $?"
tail -40 gpurun_out/r2s/probe.log
