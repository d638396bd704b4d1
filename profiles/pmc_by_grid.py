#!/usr/bin/env python3
"""Per (kernel, grid size in threads) summary of a tools/profile_cfg5.sh run: mean duration from the kernel trace,
FETCH_SIZE / WRITE_SIZE per launch from the two separate --pmc passes (rocprofv3 KiB x 1024; `fetch_bytes_x2`
is the MI355X_MICROARCH.md correction for wide 16 B/lane reads), as JSON:

    python profiles/pmc_by_grid.py gpurun_out/prof_cfg5 [name filter] > profiles/<round>_pmc_cfg5_fp16.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("sctc::", "")


def main(root, filt=""):
    out = defaultdict(dict)
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
        for k, v in acc.items():
            out[k]["launches"] = len(v)
            out[k]["avg_ms"] = sum(v) / len(v)
    for sub, ctr, key in (("pmc_fetch", "FETCH_SIZE", "fetch_bytes"), ("pmc_write", "WRITE_SIZE", "write_bytes")):
        acc = defaultdict(list)
        for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    acc[(short(r["Kernel_Name"]), r["Grid_Size"])].append(float(r["Counter_Value"]) * 1024)
        for k, v in acc.items():
            out[k][key] = sum(v) / len(v)
            if key == "fetch_bytes":
                out[k]["fetch_bytes_x2"] = 2 * out[k][key]
    rows = [dict(kernel=k[0], grid=k[1], **v) for k, v in out.items() if filt in k[0]]
    rows.sort(key=lambda r: -(r.get("avg_ms", 0) * r.get("launches", 0)))
    json.dump({"source": "tools/profile_cfg5.sh: rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE "
                         "(three separate passes) of `python tools/cfg5_step.py 8 2` (cfg-5, fp16 operands, minibatch 8)",
               "kernels": rows}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
