#!/bin/bash
# round 4, GPU session D: the new parity tests (cfg-5 gradients at size, fewer utterances than lanes)
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fp16.py tests/test_gpu_shared.py -x -q -k "cfg5 or fewer_utterances or cfg2 or cfg4_minibatch" > gpurun_out/r4d_tests.log 2>&1
tail -15 gpurun_out/r4d_tests.log
grep -i "cfg5\|cfg2\|cfg4" gpurun_out/test_notes.txt | tail -20
