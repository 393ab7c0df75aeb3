#!/usr/bin/env python3
"""avg us per launch of the kernels whose name contains one of the given needles, from a rocpd_summary.py CSV on stdin."""
import csv
import sys

needles = sys.argv[1:]
for r in csv.reader(sys.stdin):
    if len(r) >= 5 and r[0] != "kernel" and (not needles or any(n in r[0] for n in needles)):
        name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{name[:48]:48s} calls {r[1]:>5s}  avg {float(r[3]):8.2f} us  {r[4]:>6s} %")
