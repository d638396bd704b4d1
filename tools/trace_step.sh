#!/bin/bash
# kernel timeline of the last bench step: tools/trace_step.sh "ENV=.." -> gpurun_out/trace_step.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/trace_step; rm -rf $O
cd $R
env $1 rocprofv3 --kernel-trace --output-format csv -d $O -- python bench.py --steps 2 --warmup 1 --no-side --no-cpu-baseline > /tmp/trace_bench.log 2>&1
f=$(find $O -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $R/gpurun_out/trace_step.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = after the last softmax launch minus a forward pass: take the last 120 kernels
rows = rows[-130:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"]
    n = n.replace("void sctc::", "").split("(")[0][:44]
    print("%-46s q%-3s %9.3f %9.3f  grid %s" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
PY
tail -3 /tmp/trace_bench.log | cut -c1-200
