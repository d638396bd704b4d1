#!/bin/bash
# A/B of environment settings on the headline bench step (alternating runs, one line each):
#   tools/ab_env.sh ROUNDS "VAR=1 OTHER=2" "VAR=0" ...
R=$1; shift
for r in $(seq $R); do
  for e in "$@"; do
    env $e python bench.py --no-side --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phase_ms']
print('[$e]', round(d['value']), round(d['ms_per_step'],3), 'cost', d['cost_mean'], {k: round(v,3) for k,v in p.items()})"
  done
done
