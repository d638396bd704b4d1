cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in libsctc_hip.so libvar_nodiet.so; do
  echo -n "$lib: "
  SCTC_LIB_PATH=$PWD/stanford-ctc_amd/$lib timeout 300 python tools/ctc_paths_bench.py --shapes sat,sat1k,cfg2 --paths fused --reps 5 2>&1 | grep "^{" | python -c "
import sys, json
print(' | '.join('%s B=%d %.3f ms' % (d['shape'], d['B'], d['gpu_ms']) for d in map(json.loads, sys.stdin)))"
done; done
