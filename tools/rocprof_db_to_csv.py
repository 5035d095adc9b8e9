#!/usr/bin/env python3
"""Dump the kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) to a small CSV for profiles/.

    python tools/rocprof_db_to_csv.py gpurun_out/prof_r01/r01_results.db profiles/r01_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, total, avg, pct in cur:
            short = name if len(name) < 160 else name[:60] + " ... " + name[-60:]
            w.writerow([short, calls, "%.3f" % total, "%.3f" % avg, "%.3f" % pct])
    # per-dispatch detail of our kernels (grid, LDS, registers) when available
    try:
        cur = c.execute("select name, count(*), min(end-start), avg(end-start), max(end-start) from kernels "
                        "where name like '%k_step%' or name like '%k_reset%' group by name")
        with open(out.replace(".csv", "_dispatch.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "dispatches", "min_ns", "avg_ns", "max_ns"])
            for row in cur:
                w.writerow(row)
    except sqlite3.Error as e:
        print("no per-dispatch table:", e)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
