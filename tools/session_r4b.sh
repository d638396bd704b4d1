#!/bin/bash
# round 4, GPU session B: static issue priority between the two chains of a CU
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
echo "== A/B round 1"; timeout 300 python tests/gpu_ab_rec.py 1000 0,5,6,7,8,24,28,25 32
echo "== A/B round 2"; timeout 300 python tests/gpu_ab_rec.py 1000 25,28,24,8,7,6,5,0 32
} > gpurun_out/r4b_ab.log 2>&1
{
echo "== timeline variant 5"; SCTC_REC_VARIANT=5 timeout 120 python tests/gpu_diag.py recdbg1
echo "== timeline variant 24"; SCTC_REC_VARIANT=24 timeout 120 python tests/gpu_diag.py recdbg1
} > gpurun_out/r4b_timeline.log 2>&1
grep variant gpurun_out/r4b_ab.log | cut -c1-150
