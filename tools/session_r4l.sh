#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_ctc.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
timeout 300 python tests/gpu_fuzz_ctc.py 200 11 2>&1 | tail -2
timeout 300 python tests/gpu_diag.py brnn5bh brnn5h brnn 2>&1 | grep "step\|phases"
