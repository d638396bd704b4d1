#!/usr/bin/env python3
"""Condenses the passes of tools/profile_ctc_bound.sh into one JSON: per CTC kernel the LAST dispatch's counters (the
bench issues a warm-up call and one measured call per path), its duration, and the derived figures the question
"issue-bound or latency-bound?" needs:
  waves, wave_cycles (quad-cycles summed over waves), valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of a wave's
  resident time in which it has a VALU instruction executing), issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, parked =
  SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier), stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stall),
  per-wave instruction counts (VALU, of which float64 add / mul / fma, SALU, LDS, VMEM),
  simd_valu_util = SQ_ACTIVE_INST_VALU x 4 / (duration x clock x 1024 SIMDs) with the clock taken from
  SQ_BUSY_CYCLES where available.
usage: tools/ctc_bound_summary.py <dir> <shape>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, shape = sys.argv[1], sys.argv[2]
vals = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> [per dispatch]
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(lambda: defaultdict(float))     # (kernel, dispatch) -> counter -> sum over dimensions
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if "ctc" not in name and "softmax" not in name:
                continue
            per[(name, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    for (name, disp), cs in sorted(per.items(), key=lambda kv: int(kv[0][1])):
        for c, v in cs.items():
            vals[name][c].append(v)
dur = {}
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if "ctc" in name or "softmax" in name:
                dur.setdefault(name, []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
import hashlib
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stanford-ctc_amd", "csrc")
h = hashlib.sha256()
for f in ("ctc_fused.hip", "ctc_store.h", "ctc_kernels.h", "xlane.h", "common.h"):
    h.update(open(os.path.join(csrc, f), "rb").read())
out = {"shape": shape, "source_hash_ctc": h.hexdigest()[:16], "kernels": {}}
for name, cs in vals.items():
    last = {c: v[-1] for c, v in cs.items()}
    k = {"counters_last_dispatch": last, "dispatches": max(len(v) for v in cs.values())}
    if name in dur:
        k["duration_ms_last_dispatch"] = dur[name][-1] / 1e6
    wc = last.get("SQ_WAVE_CYCLES")
    waves = last.get("SQ_WAVES")
    if wc:
        for key, c in (("valu_busy", "SQ_ACTIVE_INST_VALU"), ("issue_busy", "SQ_ACTIVE_INST_ANY"), ("parked", "SQ_WAIT_ANY"),
                       ("issue_stalled", "SQ_WAIT_INST_ANY")):
            if c in last:
                k[key] = last[c] / wc
    if waves:
        k["per_wave"] = {c.replace("SQ_INSTS_", "").lower(): last[c] / waves for c in last if c.startswith("SQ_INSTS_")}
        if wc:
            k["per_wave"]["wave_quad_cycles"] = wc / waves
    if "GRBM_GUI_ACTIVE" in last and "SQ_ACTIVE_INST_VALU" in last:
        # cycles in which a SIMD executes a VALU instruction (quad-cycles x 4, summed over all waves) against the cycles
        # the device's 1024 SIMDs offer while the kernel runs (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        cyc = last["GRBM_GUI_ACTIVE"] / 8.0
        k["kernel_cycles"] = cyc
        k["simd_valu_util"] = last["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024)
        if "duration_ms_last_dispatch" in k:
            k["shader_clock_GHz_while_profiled"] = cyc / (k["duration_ms_last_dispatch"] * 1e-3) / 1e9
        f64 = sum(last.get("SQ_INSTS_VALU_%s_F64" % n, 0.0) for n in ("ADD", "MUL", "FMA", "TRANS"))
        if last.get("SQ_INSTS_VALU"):
            k["float64_share_of_valu_instructions"] = f64 / last["SQ_INSTS_VALU"]
    out["kernels"][name[:90]] = k
print(json.dumps(out, indent=1))
