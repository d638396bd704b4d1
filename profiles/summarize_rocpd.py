#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.x, rocpd sqlite output) results.db into the per-kernel
summary CSV `rocprofv3 --stats` prints: name, calls, total ns, average ns, percentage.
Usage: python profiles/summarize_rocpd.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage "
                            "from top_kernels"))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for name, calls, total, avg, pct in rows:
            if len(name) > 160:
                name = name[:157] + "..."
            w.writerow([name, calls, int(total * 1000) if total < 1e9 else int(total), "%.1f" % (avg * 1000), "%.3f" % pct])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
