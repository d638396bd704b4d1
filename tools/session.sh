#!/bin/bash
# One parametrised gpurun session script (round 5; replaces the twenty session_r4*.sh transcripts):
#   gpurun --timeout N -- 'bash tools/session.sh <case> [args]'
# Every step runs under its own `timeout`, logs go to gpurun_out/ (merged back by gpurun).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
case "$1" in
ctc)      # the three CTC paths: parity tests, then their kernels under rocprofv3 (per-kernel durations)
    timeout 900 python -m pytest tests/test_gpu_ctc_paths.py tests/test_gpu_ctc.py "tests/test_gpu_fuzz.py::test_fuzz_ctc" -x -q -s > gpurun_out/ctc_tests.log 2>&1
    tail -25 gpurun_out/ctc_tests.log
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_ctc -- python $OLDPWD/tools/ctc_paths_bench.py --shapes ${2:-cfg3,sat,cfg5} > $OLDPWD/gpurun_out/ctc_paths.log 2>&1)
    grep '^{' gpurun_out/ctc_paths.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%-5s %-8s wall %8.3f ms gpu %8.3f ms  %7.1f GB/s(gpu)  skipped %d  dcost %.1e dgrad %.1e' % (d['shape'], d['path'], d['wall_ms'], d['gpu_ms'], d['algorithmic_GBps_gpu'], d['skipped'], d['max_cost_rel_vs_first_path'], d['max_grad_abs_vs_first_path']))"
    f=$(find gpurun_out/prof_ctc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "ctc\|softmax\|Name" $f | cut -c1-200 | head -30
    find gpurun_out/prof_ctc -name "*kernel_trace.csv" -size +5M -delete
    ;;
bench)    # the headline line without the CPU leg
    shift
    timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err
    tail -3 gpurun_out/bench.err
    python tools/bench_digest.py gpurun_out/bench.json
    ;;
tests)    # pytest -m gpu over the given files / node ids (default: everything)
    shift
    rm -f gpurun_out/test_notes.txt
    timeout 2400 python -m pytest ${@:-tests} -m gpu -x -q > gpurun_out/suite.log 2>&1
    tail -12 gpurun_out/suite.log
    ;;
final)    # round-end evidence: full suite, smoke, the rocprofv3 passes of bench.py, the default bench line
    TAG=${2:-r05}
    rm -f gpurun_out/test_notes.txt
    timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1
    tail -6 gpurun_out/suite.log
    timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
    timeout 1500 bash tools/profile_bench.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
    tail -3 gpurun_out/profile_$TAG.log | cut -c1-200
    cp gpurun_out/prof_$TAG/pmc_summary.json profiles/${TAG}_pmc_summary.json   # bench.py reads the newest summary
    timeout 1200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
    python tools/bench_digest.py gpurun_out/bench_default.json
    ;;
*)
    echo "usage: tools/session.sh ctc|bench|tests|final [args]"; exit 2;;
esac
