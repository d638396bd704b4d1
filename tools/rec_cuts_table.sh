#!/bin/bash
# the numbers behind profiles/r06_recurrence_2d.md's tables: us per time step of the recurrence at H = 1824 for
# 40..128 utterances -- default dispatch (0), one launch whatever the count (45), round 5's one-slab-per-CU kernel (47)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
for B in ${BS:-40 48 64 72 80 96 112 128}; do
  timeout 300 python tools/rec_variant_time.py $B 0 45 47 2>&1 | grep "^B=" | tail -3
done
echo "== ragged U[T/2, T], default dispatch against round 5's kernel"
RAGGED=1 CFGS=0 BS="${BS:-40 48 64 72 80 96 112 128}" bash tools/rec_tiled_sweep.sh
echo "== H = 2048"
H=2048 CFGS=0 BS="64 128" bash tools/rec_tiled_sweep.sh
