#!/bin/bash
# rocprofv3 evidence for the cfg-5 fp16 step (T=8000, 7x2048, minibatch 8): kernel trace + FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes (one counter group per pass, each under its own timeout; see profile_bench.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cfg5
rm -rf $O; mkdir -p $O
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/cfg5_step.py 8 3 > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python tools/cfg5_step.py 8 2 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python tools/cfg5_step.py 8 2 > $O/write.log 2>&1
tail -1 $O/stats.log
python profiles/pmc_by_grid.py $O > $O/pmc_by_grid.json
find $O -name "*.csv" -size +8M -delete
python - <<EOF
import json
d=json.load(open('$O/pmc_by_grid.json'))
for r in d['kernels'][:14]:
    print("%-46s %-12s n=%3d %.3f ms fetch %.0f MB write %.0f MB" % (r['kernel'][:46], r['grid'], r.get('launches',0), r.get('avg_ms',0), r.get('fetch_bytes',0)/1e6, r.get('write_bytes',0)/1e6))
EOF
