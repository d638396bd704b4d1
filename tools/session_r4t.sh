#!/bin/bash
# ctc_grad with host-built per-label state lists: parity tests, then its duration in the cfg-5 and cfg-3 steps
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
{
timeout 900 python -m pytest tests/test_gpu_ctc.py tests/test_gpu_fuzz.py -q -x -m gpu 2>&1 | tail -8
timeout 300 python tests/gpu_fuzz_ctc.py 60 3 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5 -- python $R/tools/cfg5_step.py 8 3 > /tmp/c5.log 2>&1
tail -1 /tmp/c5.log; grep "ctc_\|softmax" /tmp/c5/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
rm -rf /tmp/c3; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3 -- python $R/bench.py --no-side --no-cpu-baseline --steps 6 --warmup 2 > /tmp/c3.log 2>&1
tail -1 /tmp/c3.log | cut -c1-200; grep "ctc_\|softmax" /tmp/c3/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
} > $R/gpurun_out/r4t.log 2>&1
tail -40 $R/gpurun_out/r4t.log
