#!/usr/bin/env python3
"""Condenses the rocprofv3 passes written by tools/profile_bench.sh into one small JSON:

    python profiles/pmc_to_json.py gpurun_out/prof_<tag> > profiles/<round>_pmc_summary.json

Per kernel family (all launches of `bench.py --no-side`, i.e. full-size cfg-3 steps only):
launches per costAndGrad, mean duration, FETCH_SIZE / WRITE_SIZE per launch (rocprofv3 reports
KiB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE shows half the bytes of wide coalesced 16 B/lane
reads, so `fetch_bytes_x2` is the corrected figure for such kernels; WRITE_SIZE is uncalibrated),
and MFMA utilisation by the gfx94x formula
    MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE[sum over 8 XCDs] * 32 CUs * 4 SIMDs).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FAMILIES = ["gemm_f32_kernel", "brnn_recurrent", "ctc_fused_kernel", "ctc_lattice_kernel", "ctc_grad_kernel",
            "softmax_rows_kernel", "splitk_reduce_kernel", "colsum_partial_kernel",
            "colsum_final_kernel", "add_kernel", "gather_rows_kernel", "scatter_rows_kernel"]


def family(name):
    for f in FAMILIES:
        if f in name:
            return f
    return None


def read_pass(d):
    """{family: {counter: [values]}} , {family: [durations ns]}"""
    vals = defaultdict(lambda: defaultdict(list))
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                fam = family(row["Kernel_Name"])
                if fam:
                    vals[fam][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return vals


def read_stats(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                fam = family(row["Name"])
                if fam:
                    o = out.setdefault(fam, {"calls": 0, "total_ns": 0.0})
                    o["calls"] += int(row["Calls"])
                    o["total_ns"] += float(row["TotalDurationNs"])
    return out


def main(root):
    mf = read_pass(os.path.join(root, "pmc_mfma"))
    fe = read_pass(os.path.join(root, "pmc_fetch"))
    wr = read_pass(os.path.join(root, "pmc_write"))
    st = read_stats(os.path.join(root, "stats"))
    out = {"source": "rocprofv3 --kernel-trace --stats and three separate --pmc passes of "
                     "`python bench.py --no-side --no-cpu-baseline` (tools/profile_bench.sh)",
           "units": {"fetch/write": "bytes per launch (rocprofv3 KiB x 1024)"}, "kernels": {}}
    # one softmax_rows launch per costAndGrad (round 5: the CTC kernel of a step is ctc_fused_kernel OR the
    # ctc_lattice / ctc_grad pair, the softmax is always there)
    steps_stats = st.get("softmax_rows_kernel", {}).get("calls", 0)
    steps_pmc = len(mf.get("softmax_rows_kernel", {}).get("SQ_WAVES", [])) or \
        len(fe.get("softmax_rows_kernel", {}).get("FETCH_SIZE", []))
    out["costAndGrad_calls"] = {"stats_pass": steps_stats, "pmc_passes": steps_pmc}
    if len(sys.argv) > 2:
        out["source_hash"] = sys.argv[2]      # bench.csrc_hash() of the measured tree
    mean = lambda v: sum(v) / len(v) if v else None
    for fam in FAMILIES:
        k = {}
        if fam in st and steps_stats:
            k["launches_per_step"] = st[fam]["calls"] / steps_stats
            k["avg_launch_ms"] = st[fam]["total_ns"] / st[fam]["calls"] * 1e-6
            k["ms_per_step"] = st[fam]["total_ns"] / steps_stats * 1e-6
        if fam in fe:
            f = mean(fe[fam]["FETCH_SIZE"])
            k["fetch_bytes"] = f * 1024
            k["fetch_bytes_x2"] = 2 * f * 1024
        if fam in wr:
            k["write_bytes"] = mean(wr[fam]["WRITE_SIZE"]) * 1024
        if fam in mf and mf[fam].get("GRBM_GUI_ACTIVE"):
            busy, act = sum(mf[fam]["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(mf[fam]["GRBM_GUI_ACTIVE"])
            k["mfma_busy_cycles_per_launch"] = mean(mf[fam]["SQ_VALU_MFMA_BUSY_CYCLES"])
            k["gui_active_cycles_per_launch_sum_xcd"] = mean(mf[fam]["GRBM_GUI_ACTIVE"])
            k["mfma_util"] = busy / (act * 32 * 4)
        if k:
            out["kernels"][fam] = k
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
