#!/usr/bin/env python3
"""Groups a rocprofv3 --pmc counter_collection.csv by kernel name: mean counter value per
dispatch.  Usage: python profiles/pmc_summary.py <dir-with-csvs> [substring filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, filt=""):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if filt and filt not in name:
                    continue
                key = name[:70] + " grid=" + row.get("Grid_Size", "?")
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c, vals in sorted(acc[k].items()):
            print("    %-32s n=%-4d mean=%.4g" % (c, len(vals), sum(vals) / len(vals)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
