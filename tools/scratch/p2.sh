cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cfg2
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python tests/gpu_diag.py brnn2 > $O/log.txt 2>&1
grep -A2 "step " $O/log.txt | head -4
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_cfg2/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s calls %4s avg %8.1f us total %8.2f ms" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
