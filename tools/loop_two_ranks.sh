#!/bin/bash
# VERDICT r02 #1 "done" criterion: the 2-rank bench rehearsal on ONE GPU (both ranks launch
# whole-device persistent recurrent grids; round 2 dead-locked here) N times in a row.
# usage: tools/loop_two_ranks.sh [N=20] [log=gpurun_out/r03_two_ranks_loop.log]
N=${1:-20}
LOG=${2:-gpurun_out/r03_two_ranks_loop.log}
mkdir -p "$(dirname "$LOG")"
: > "$LOG"
export SCTC_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
pass=0
for i in $(seq 1 "$N"); do
  t0=$(date +%s%N)
  out=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port $((29600 + i)) bench.py --gpus 2 --steps 2 --warmup 1 --batch 6 --no-side --no-cpu-baseline 2>&1)
  rc=$?
  t1=$(date +%s%N)
  line=$(echo "$out" | grep '^{' | tail -1)
  val=$(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read() or "{}"); print("%.0f frames/s, cost rel err %.1e" % (d.get("value",0), d.get("cost_check",{}).get("rel_err",-1)))' 2>/dev/null)
  if [ $rc -eq 0 ] && [ -n "$line" ]; then pass=$((pass+1)); st=PASS; else st=FAIL; echo "$out" | tail -20 >> "$LOG"; fi
  printf "run %2d: %s rc=%d %.1fs %s\n" "$i" "$st" "$rc" "$(( (t1 - t0) / 100000000 ))e-1" "$val" | tee -a "$LOG"
done
echo "two-rank bench on one GPU: $pass/$N passed" | tee -a "$LOG"
[ "$pass" -eq "$N" ]
