#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of the lattice + grad CTC path per shape of tools/ctc_paths_bench.py
#   gpurun -- 'bash tools/lattice_path_times.sh [shapes]'      env SCTC_LIB_PATH selects a variant library
cd /tmp && export TMPDIR=/tmp
for sh in ${1:-cfg5 cfg3 sat cfg4}; do
  rm -rf /tmp/pg; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python $GRAFT_REPO_ROOT/tools/ctc_paths_bench.py --shapes $sh --paths lattice --reps 3 > /tmp/pg.log 2>&1
  f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1); echo "== $sh"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ctc_grad" in r["Name"] or "ctc_lattice" in r["Name"]:
        print("  %-60s calls %3s avg %9.1f us min %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
