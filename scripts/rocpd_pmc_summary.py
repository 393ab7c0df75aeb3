#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database, as CSV.
Run on the GPU box right after `rocprofv3 --pmc ... --kernel-trace` (the .db files are too big
to ship back):   python scripts/rocpd_pmc_summary.py /tmp/x/pmc_results.db > gpurun_out/pmc.csv
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def main(path):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    sys.stderr.write("counters_collection columns: %s\n" % cols)
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    q = f"select {name_col}, counter_name, value, dispatch_id from counters_collection"
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for kname, cname, val, did in con.execute(q):
        agg[kname][cname] += float(val)
        calls[kname].add(did)
    counters = sorted({c for k in agg for c in agg[k]})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches"] + [c + "_avg" for c in counters])
    dur = {}
    try:
        for n, c, t in con.execute("select name, count(*), sum(end-start) from kernels group by name"):
            dur[n] = (c, t)
    except sqlite3.Error:
        pass
    rows = []
    for k in agg:
        n = max(len(calls[k]), 1)
        rows.append((dur.get(k, (0, 0))[1], [k[:110], n] + [round(agg[k][c] / n, 1) for c in counters]))
    for _, r in sorted(rows, key=lambda x: -x[0])[:40]:
        w.writerow(r)


if __name__ == "__main__":
    main(sys.argv[1])
