#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (tools/pmc_profile.sh) for kernels matching a name: per-dispatch averages.

    python tools/pmc_summary.py gpurun_out/pmc_r01 k_step > profiles/r01_pmc_summary.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, pat):
    acc = defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat in row.get("Kernel_Name", ""):
                    a = acc[row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    for k in sorted(acc):
        s, n = acc[k]
        print("%-28s avg/dispatch %16.1f   dispatches %d" % (k, s / n, n))
    dur = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat in row.get("Kernel_Name", ""):
                    dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    if dur:
        print("kernel duration under PMC: avg %.1f us over %d dispatches" % (sum(dur) / len(dur) / 1e3, len(dur)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_step")
