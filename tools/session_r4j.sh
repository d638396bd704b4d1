#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/prof_cfg5
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_cfg5 -- python /root/repo/tools/cfg5_step.py 8 3 > /root/repo/gpurun_out/prof_cfg5/run.log 2>&1
cd /root/repo
ls -R gpurun_out/prof_cfg5 | head -20
f=$(find gpurun_out/prof_cfg5 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-90s calls %5s total %9.3f ms avg %9.3f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
