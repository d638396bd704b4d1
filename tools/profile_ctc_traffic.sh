#!/bin/bash
# HBM traffic of the CTC kernels ALONE at a batch that cannot stay in the L2s (default: the saturating batch, 4096
# utterances of the cfg-3 shape): two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never in one pass,
# never with a trace domain) over tools/ctc_paths_bench.py, summarised per kernel (KiB per dispatch).
#   gpurun --timeout 600 -- 'bash tools/profile_ctc_traffic.sh [shape] [paths]'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-sat}
PATHS=${2:-fused,lattice}
O=$R/gpurun_out/pmc_ctc_$SHAPE
rm -rf $O; mkdir -p $O
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python tools/ctc_paths_bench.py --shapes $SHAPE --paths $PATHS --reps 1 > $O/run_$c.log 2>&1
    python profiles/pmc_summary.py $O/$c ctc > $O/${c}_summary.txt 2>&1
    find $O/$c -name "*counter_collection.csv" -size +5M -delete
done
cat $O/FETCH_SIZE_summary.txt $O/WRITE_SIZE_summary.txt
grep '^{' $O/run_FETCH_SIZE.log | cut -c1-200
