#!/bin/bash
# full GPU suite + headline bench (no CPU leg), logs to gpurun_out/
cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rm -f gpurun_out/test_notes.txt
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1
tail -8 gpurun_out/suite.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print("value %.0f  ms %.3f  hbm_resident %.0f  phases %s" % (d["value"], d["ms_per_step"], d["hbm_resident"]["value"], {k: round(v, 2) for k, v in d["phase_ms"].items()}))
print("rec by minibatch", {k: (round(v["us_per_time_step"], 2), round(v["frac_of_f32_mfma_peak"], 3)) for k, v in d["roofline_recurrent"]["by_minibatch"].items()})
c5 = d["cfg5_fp16"]
for k in ("minibatch_8", "minibatch_1"):
    print(k, round(c5[k]["value"]), round(c5[k]["ms_per_step"], 2), "gemm frac", round(c5[k]["roofline_gemm"]["frac"], 3), {a: round(b, 2) for a, b in c5[k]["phase_ms"].items()})
print("streams", d["one_utterance_per_stream"]["value"], d["one_utterance_per_stream"]["streams"])
PY
