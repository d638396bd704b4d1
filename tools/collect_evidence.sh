#!/bin/bash
# (kernel stats: the NEWEST file -- gpurun merges every session's run directory into gpurun_out/)
# copies what `tools/session.sh final <tag>` (+ profile_cfg5.sh, `session.sh ctc`) left in gpurun_out/ into profiles/<tag>_*
TAG=${1:-r05}
cd "$(dirname "$0")/.."
f=$(ls -t $(find gpurun_out/prof_$TAG/stats -name "*kernel_stats.csv") | head -1) && cp $f profiles/${TAG}_bench_kernel_stats.csv
cp gpurun_out/prof_$TAG/pmc_summary.json profiles/${TAG}_pmc_summary.json
python - <<PY
import json
d = json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
json.dump(d, open('profiles/${TAG}_bench.json', 'w'), indent=1)
print("value %.0f ms %.3f frac %.4f" % (d['value'], d['ms_per_step'], d['roofline']['frac']))
PY
cp gpurun_out/suite.log profiles/${TAG}_pytest_gpu_full.log
cp gpurun_out/test_notes.txt profiles/${TAG}_test_notes.txt
f=$(ls -t $(find gpurun_out/prof_cfg5/stats -name "*kernel_stats.csv") | head -1) && cp $f profiles/${TAG}_cfg5_fp16_kernel_stats.csv
cp gpurun_out/prof_cfg5/pmc_by_grid.json profiles/${TAG}_pmc_cfg5_fp16.json
f=$(ls -t $(find gpurun_out/prof_ctc -name "*kernel_stats.csv") | head -1) && cp $f profiles/${TAG}_ctc_paths_kernel_stats.csv
grep "^{" gpurun_out/ctc_paths.log > profiles/${TAG}_ctc_paths.jsonl
python -c "import bench, json; print('tree', bench.csrc_hash(), 'summary', json.load(open('profiles/${TAG}_pmc_summary.json'))['source_hash'])"
