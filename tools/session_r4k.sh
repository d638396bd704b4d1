#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
for pd in -1 0 2 8; do
echo "== poll delay $pd"; SCTC_REC_POLL_DELAY=$pd timeout 300 python tests/gpu_diag.py brnn5bh 2>&1 | grep "step\|phases"
done
} > gpurun_out/r4k_mh.log 2>&1
cat gpurun_out/r4k_mh.log
timeout 900 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
SCTC_FUZZ_FP16=1 timeout 300 python tests/gpu_fuzz.py 10 7 2>&1 | tail -2
