#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
export SCTC_G16_NO_CHILD=1
{
for v in base g16a128 g16a32; do
  if [ "$v" = base ]; then unset SCTC_LIB_PATH; else export SCTC_LIB_PATH=/root/repo/stanford-ctc_amd/libvar_$v.so; fi
  echo "== $v"; timeout 300 python tests/gpu_g16.py speed 2>&1 | grep "bf16" | grep "fwd / dgrad\|square 8k\|input\|fwd B=1"
done
} > gpurun_out/r4f_ablate.log 2>&1
cat gpurun_out/r4f_ablate.log
