#!/bin/bash
# A/B of kernel-variant libraries on the headline bench step (alternating runs, one line each):
#   tools/ab_bench.sh ROUNDS libA.so libB.so ...      ("base" = the shipped libsctc_hip.so)
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    if [ "$lib" = base ]; then unset SCTC_LIB_PATH; else export SCTC_LIB_PATH=/root/repo/stanford-ctc_amd/$lib; fi
    python bench.py --no-side --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phase_ms']
print('$lib', round(d['value']), round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4), {k: round(v,3) for k,v in p.items()})"
  done
done
