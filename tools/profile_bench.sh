#!/bin/bash
# rocprofv3 evidence for bench.py: kernel-trace stats pass + three separate PMC passes, each under its own
# `timeout` (round 4: FETCH_SIZE and WRITE_SIZE in ONE pass exceed the counter hardware -- rocprofv3 aborts
# and then hangs in its finalizer: 30 GPU-minutes lost; one counter group per pass, always)
# (never combined with sys/runtime/hip trace domains).  Run on the GPU box through gpurun:
#   gpurun -- 'bash tools/profile_bench.sh r01_final'
# then copy gpurun_out/prof_<tag>/{stats/*kernel_stats.csv,pmc_summary.json} into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-side --no-cpu-baseline > $O/bench_stats.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_mfma -- python bench.py --steps 2 --warmup 1 --no-side --no-cpu-baseline > $O/bench_pmc_mfma.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-side --no-cpu-baseline > $O/bench_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 2 --warmup 1 --no-side --no-cpu-baseline > $O/bench_pmc_write.log 2>&1
find $O -name "*.csv" | head -30
du -sh $O
tail -2 $O/bench_stats.log | cut -c1-300
for f in mfma fetch write; do python profiles/pmc_summary.py $O/pmc_$f > $O/pmc_${f}_summary.txt 2>&1; done
# keep only small files
find $O -name "*counter_collection.csv" -size +20M -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
python profiles/pmc_to_json.py $O $(python -c "import bench; print(bench.csrc_hash())") > $O/pmc_summary.json
head -c 1500 $O/pmc_summary.json
