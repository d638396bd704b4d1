#!/usr/bin/env python3
"""prints the fields of a bench.py JSON line that a session looks at first"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f %s  ms %.3f  frac %.4f  traffic %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic")))
print("phases", {k: round(v, 3) for k, v in d.get("phase_ms", {}).items()})
rc = d.get("roofline_ctc", {})
print("ctc", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rc.items() if k in ("achieved", "frac", "ms", "traffic", "traffic_ratio", "algorithmic_bytes")})
if "long_rows" in rc:
    print("ctc long rows", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in rc["long_rows"].items() if k != "note"})
if "saturating_batch" in rc:
    print("ctc saturating", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in rc["saturating_batch"].items() if k != "note"})
if "hbm_resident" in d:
    print("hbm_resident %.0f" % d["hbm_resident"]["value"])
rr = d.get("roofline_recurrent", {})
if "minibatch_128" in d:
    print("minibatch_128", round(d["minibatch_128"]["value"]), "frames/s", round(d["minibatch_128"]["ms_per_step"], 1), "ms")
if "by_minibatch" in rr:
    print("rec by minibatch", {k: (round(v["us_per_time_step"], 2), round(v["frac_of_f32_mfma_peak"], 3)) for k, v in rr["by_minibatch"].items()})
for k in ("cfg1_minibatch1", "cfg2_minibatch1"):
    if k in d:
        print(k, round(d[k]["value"]), "frames/s", round(d[k]["ms_per_utterance"], 3), "ms", {a: round(b, 3) for a, b in d[k]["phase_ms"].items()})
c4 = d.get("cfg4_share")
if c4:
    print("cfg4 share", round(c4["value"]), round(c4["ms_per_step"], 2), {a: round(b, 2) for a, b in c4["phase_ms"].items()})
c5 = d.get("cfg5_fp16")
if c5:
    for k in ("minibatch_8", "minibatch_1"):
        print(k, round(c5[k]["value"]), round(c5[k]["ms_per_step"], 2), {a: round(b, 2) for a, b in c5[k]["phase_ms"].items()})
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
