#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
export SCTC_G16_NO_CHILD=1
{
for pad in 0 64 192 32; do
  echo "== pad $pad"; SCTC_G16_PAD=$pad timeout 300 python tests/gpu_g16.py speed 2>&1 | grep "bf16" | grep "fwd / dgrad\|wgrad  \|square 8k\|square 4k\|input"
done
} > gpurun_out/r4g_pad.log 2>&1
cat gpurun_out/r4g_pad.log
