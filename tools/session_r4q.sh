#!/bin/bash
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout 200 python tools/gemm_context_probe.py > gpurun_out/r4q_probe.log 2>&1
tail -40 gpurun_out/r4q_probe.log
