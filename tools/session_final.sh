#!/bin/bash
# round-end evidence: full GPU suite, the default bench line (with the CPU leg), the rocprofv3 passes
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rm -f gpurun_out/test_notes.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1
tail -6 gpurun_out/suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
timeout 1300 bash tools/profile_bench.sh r04 > gpurun_out/profile_r04.log 2>&1
tail -3 gpurun_out/profile_r04.log | cut -c1-200
# the bench line's roofline.traffic comes from the committed summary of the SAME sources: use the one just measured
cp gpurun_out/prof_r04/pmc_summary.json profiles/r04_pmc_summary.json
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value %.0f  ms %.3f  hbm_resident %.0f  frac %.4f traffic %s phases %s" % (d["value"], d["ms_per_step"], d["hbm_resident"]["value"], d["roofline"]["frac"], d["roofline"]["traffic"], {k: round(v, 2) for k, v in d["phase_ms"].items()}))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["single_thread"]["value"])
c5 = d["cfg5_fp16"]
for k in ("minibatch_8", "minibatch_1"):
    print(k, round(c5[k]["value"]), round(c5[k]["ms_per_step"], 2), "gemm frac", round(c5[k]["roofline_gemm"]["frac"], 3))
PY
