#!/bin/bash
# A/B of two builds of the library on the CTC paths bench (alternating, three rounds):
#   tools/ab_ctc_libs.sh libA.so libB.so [shapes]      (files under stanford-ctc_amd/; tools/build_variant.sh makes variants)
cd ${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; SH=${3:-sat,sat1k,cfg2,cfg3,cfg4}
for rep in 1 2 3; do
for lib in $A $B; do
  echo -n "$lib: "
  SCTC_LIB_PATH=$PWD/stanford-ctc_amd/$lib timeout 300 python tools/ctc_paths_bench.py --shapes $SH --paths fused --reps 5 2>&1 | grep "^{" | python -c "
import sys, json
print(' | '.join('%s B=%d %.3f ms' % (d['shape'], d['B'], d['gpu_ms']) for d in map(json.loads, sys.stdin)))"
done; done
