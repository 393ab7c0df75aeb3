#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (trace_results.db) into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` reports, as CSV (name, calls, total_us, avg_us, pct).

    python scripts/rocpd_summary.py gpurun_out/prof/trace_results.db > profiles/rNN_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(path, top=60):
    con = sqlite3.connect(path)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                       "order by total_duration desc").fetchall()
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
    for name, calls, total, avg, pct in rows[:top]:
        w.writerow([name[:160], calls, round(total, 1), round(avg, 2), round(pct, 2)])
    try:
        rows = con.execute("select count(*), sum(size), sum(end - start) from memory_copies").fetchone()
        w.writerow(["# memory_copies: count, bytes, total_ns"] + list(rows))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
