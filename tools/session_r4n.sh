#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 200 python tests/gpu_fuzz_gemm.py 100 3 2>&1 | tail -2
timeout 200 python tests/gpu_diag.py gemm 2>&1 | grep "^gemm"
for i in 1 2; do timeout 200 python bench.py --no-side --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phase_ms']
print(round(d['value']), round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4), {k: round(v,3) for k,v in p.items()})"; done
