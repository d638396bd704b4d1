#!/bin/bash
# round 4, GPU session A: the 8-wave both-chains recurrent kernel (variants 8..17) against the shipped two-chain kernel
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
echo "== A/B round 1"; timeout 300 python tests/gpu_ab_rec.py 1000 0,8,9,10,12,13,16,17 32
echo "== A/B round 2"; timeout 300 python tests/gpu_ab_rec.py 1000 17,16,13,12,10,9,8,0 32
echo "== ragged / short A/B (T=333)"; timeout 200 python tests/gpu_ab_rec.py 333 0,8,9 27
} > gpurun_out/r4a_ab.log 2>&1
{
echo "== fuzz, kernel under test = variant 9"; SCTC_FUZZ_VARIANT=9 timeout 400 python tests/gpu_fuzz.py 60 3
echo "== fuzz, kernel under test = variant 16"; SCTC_FUZZ_VARIANT=16 timeout 300 python tests/gpu_fuzz.py 30 4
} > gpurun_out/r4a_fuzz.log 2>&1
SCTC_REC_VARIANT=9 timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "cfg3" > gpurun_out/r4a_fullsize.log 2>&1
{
echo "== timeline variant 0"; SCTC_REC_VARIANT=0 timeout 120 python tests/gpu_diag.py recdbg1
echo "== timeline variant 8"; SCTC_REC_VARIANT=8 timeout 120 python tests/gpu_diag.py recdbg1
echo "== timeline variant 9"; SCTC_REC_VARIANT=9 timeout 120 python tests/gpu_diag.py recdbg1
} > gpurun_out/r4a_timeline.log 2>&1
tail -30 gpurun_out/r4a_ab.log; tail -5 gpurun_out/r4a_fuzz.log; tail -5 gpurun_out/r4a_fullsize.log
