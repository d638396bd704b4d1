#!/bin/bash
# Which bound does the fused CTC kernel sit on?    Separate rocprofv3 --pmc passes (SQ counters only, never
# with a trace domain) over tools/ctc_paths_bench.py at the saturating batch, summarised per kernel into one JSON:
#   gpurun --timeout 900 -- 'bash tools/profile_ctc_bound.sh [shape] [paths] [tag]'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-sat}
PATHS=${2:-fused}
TAG=${3:-r06}
O=$R/gpurun_out/pmc_ctc_bound_$SHAPE
rm -rf $O; mkdir -p $O
cd $R
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
P3="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32"
P4="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INSTS_BRANCH"
n=0
for set in "$P1" "$P2" "$P3" "$P4"; do
    n=$((n+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pass$n -- python tools/ctc_paths_bench.py --shapes $SHAPE --paths $PATHS --reps 1 > $O/run_pass$n.log 2>&1
    tail -2 $O/run_pass$n.log | cut -c1-200
done
# kernel durations of the same command (no counters)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/ctc_paths_bench.py --shapes $SHAPE --paths $PATHS --reps 1 > $O/run_stats.log 2>&1
python tools/ctc_bound_summary.py $O $SHAPE > $R/gpurun_out/${TAG}_ctc_bound_$SHAPE.json
cat $R/gpurun_out/${TAG}_ctc_bound_$SHAPE.json
find $O -name "*counter_collection.csv" -size +5M -delete
find $O -name "*kernel_trace.csv" -size +5M -delete
