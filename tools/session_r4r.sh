#!/bin/bash
# fused sums around the temporal layer: bit-identity tests, then kernel stats of the fused step
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
timeout 600 python -m pytest tests/test_gpu_brnn.py -q -x -m gpu -k "fused_into_gemms" 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
rm -rf /tmp/st$f
SCTC_FUSE_ADD=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st$f -- python $GRAFT_REPO_ROOT/bench.py --no-side --no-cpu-baseline --steps 6 --warmup 2 > /tmp/st$f.log 2>&1
echo "== SCTC_FUSE_ADD=$f"; tail -1 /tmp/st$f.log | cut -c1-330
python - <<EOF
import csv,glob,re,collections
f=glob.glob('/tmp/st$f/*/*kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=re.sub(r'\(.*','',r['Kernel_Name']).replace('void sctc::','').replace('sctc::','')
    d[(n,r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if 'gemm_f32' in k[0] or 'add' in k[0]: print("%-50s grid %-8s n=%3d avg %.3f ms"%(k[0][:50],k[1],len(v),sum(v)/len(v)))
EOF
done
} > $GRAFT_REPO_ROOT/gpurun_out/r4r.log 2>&1
tail -50 $GRAFT_REPO_ROOT/gpurun_out/r4r.log
