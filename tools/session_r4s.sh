#!/bin/bash
# fused sums: where the second addend's loads are issued (variant libraries), per-kernel durations
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
{
cd /tmp && export TMPDIR=/tmp
for v in base s2slot1 s2slot2; do
rm -rf /tmp/st$v
L=$R/stanford-ctc_amd/libvar_$v.so; [ $v = base ] && L=$R/stanford-ctc_amd/libsctc_hip.so
SCTC_LIB_PATH=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st$v -- python $R/bench.py --no-side --no-cpu-baseline --steps 6 --warmup 2 > /tmp/st$v.log 2>&1
echo "== $v"; tail -1 /tmp/st$v.log | cut -c1-330
python - <<EOF
import csv,glob,re,collections
f=glob.glob('/tmp/st$v/*/*kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=re.sub(r'\(.*','',r['Kernel_Name']).replace('void sctc::','').replace('sctc::','')
    d[(n,r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if 'true>' in k[0]: print("%-50s grid %-8s n=%3d avg %.3f ms"%(k[0][:50],k[1],len(v),sum(v)/len(v)))
EOF
done
} > $R/gpurun_out/r4s.log 2>&1
tail -30 $R/gpurun_out/r4s.log
