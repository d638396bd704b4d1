#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python tests/gpu_g16.py check > gpurun_out/r4e_g16_check.log 2>&1
tail -30 gpurun_out/r4e_g16_check.log
timeout 600 python tests/gpu_g16.py speed > gpurun_out/r4e_g16_speed.log 2>&1
cat gpurun_out/r4e_g16_speed.log | grep -v amdgpu.ids
