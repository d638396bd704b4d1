#!/bin/bash
# fp32 weight-gradient GEMM: layout x shape probe, plus OCC / BK variants (sparse operands only)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
timeout 200 python tools/gemm_layout_probe.py base
for v in occ3 occ2 bk32; do
  SCTC_LIB_PATH=$PWD/stanford-ctc_amd/libvar_$v.so timeout 120 python tools/gemm_layout_probe.py $v 1
done
} > gpurun_out/r4p_probe.log 2>&1
tail -60 gpurun_out/r4p_probe.log
