#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_fp16.py -x -q > gpurun_out/r4h_fp16_tests.log 2>&1
tail -5 gpurun_out/r4h_fp16_tests.log
{
echo "== g16"; timeout 300 python tests/gpu_diag.py brnn5bh brnn5h
echo "== x16"; SCTC_G16=0 timeout 300 python tests/gpu_diag.py brnn5bh brnn5h
} > gpurun_out/r4h_cfg5.log 2>&1
grep -v amdgpu.ids gpurun_out/r4h_cfg5.log
