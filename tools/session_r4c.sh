#!/bin/bash
# round 4, GPU session C: chain-level tokens (MFMA burst / exchange-load issue) in the 8-wave kernel
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
echo "== A/B round 1"; timeout 300 python tests/gpu_ab_rec.py 1000 0,8,9,10,11,13,15,27 32
echo "== A/B round 2"; timeout 300 python tests/gpu_ab_rec.py 1000 27,15,13,11,10,9,8,0 32
} > gpurun_out/r4c_ab.log 2>&1
grep variant gpurun_out/r4c_ab.log | cut -c1-150
{
echo "== timeline variant 9"; SCTC_REC_VARIANT=9 timeout 120 python tests/gpu_diag.py recdbg1
echo "== timeline variant 11"; SCTC_REC_VARIANT=11 timeout 120 python tests/gpu_diag.py recdbg1
} > gpurun_out/r4c_timeline.log 2>&1
{
echo "== fuzz, kernel under test = variant 11"; SCTC_FUZZ_VARIANT=11 timeout 400 python tests/gpu_fuzz.py 40 5
} > gpurun_out/r4c_fuzz.log 2>&1
tail -3 gpurun_out/r4c_fuzz.log
