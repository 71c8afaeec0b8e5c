#!/usr/bin/env python3
"""Median duration per kernel (and the median idle gap in front of it) from a rocprofv3 --kernel-trace CSV:
python tools/trace_medians.py <dir-or-csv>"""
import csv, glob, os, sys
import numpy as np
path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("asam::", "")))
rows.sort()
dur, gap = {}, {}
for i, (s, e, n) in enumerate(rows):
    dur.setdefault(n, []).append((e - s) / 1e3)
    if i: gap.setdefault(n, []).append((s - rows[i - 1][1]) / 1e3)
print(f"{'kernel':34s} {'calls':>7s} {'median us':>10s} {'p10':>8s} {'p90':>8s} {'median gap before (us)':>24s}")
for n in sorted(dur, key=lambda k: -len(dur[k])):
    d = np.array(dur[n]); g = np.array(gap.get(n, [0.0]))
    print(f"{n[:34]:34s} {len(d):7d} {np.median(d):10.2f} {np.percentile(d, 10):8.2f} {np.percentile(d, 90):8.2f} {np.median(g):24.2f}")
