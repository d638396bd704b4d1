#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1700 python -m pytest tests/test_gpu_shared.py tests/test_gpu_run.py -x -q > gpurun_out/r4i_shared.log 2>&1
tail -8 gpurun_out/r4i_shared.log
