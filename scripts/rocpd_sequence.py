#!/usr/bin/env python3
"""Ordered kernel timeline of the tail of a rocprofv3 rocpd database: name, duration and the gap
to the previous dispatch -- shows what one (graph-replayed) step is made of.

    python scripts/rocpd_sequence.py trace_results.db [n_last=90]
"""
import sqlite3
import sys


def main(path, n_last=90):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = con.execute(f"select {name}, start, end from kernels order by start").fetchall()
    rows = rows[-n_last:]
    prev = None
    for nm, s, e in rows:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        print(f"{(e - s) / 1e3:9.1f} us  gap {gap:8.1f}  {nm[:110]}")
        prev = e
    print(f"# span {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us, busy {sum(e - s for _, s, e in rows) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 90)
