#!/usr/bin/env python3
"""Host-side phases of the incremental steps of the M3500 demo, by class of step (fronts regenerated):
APRILSAM_AMD_INC_PROFILE=1 APRILSAM_AMD_INC_PROFILE_DUMP=/tmp/inc.bin python tools/inc_demo.py 3500 && python tools/inc_phases.py /tmp/inc.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], np.float32).reshape(-1, 8)
names = ["pack", "model", "upload", "plan+enqueue", "wait", "writeback", "total"]
print(f"{len(a)} incremental steps; per class: steps | MEDIAN us of " + " ".join(names) + " | MEAN total")
for lo, hi in ((-1, 0), (1, 2), (2, 4), (4, 7), (7, 10), (10, 14), (14, 1000)):
    s = a[(a[:, 7] >= lo) & (a[:, 7] < hi)]
    if len(s):
        print(f"  regenerated {lo:3d}..{hi - 1:<4d} {len(s):5d} | " + " ".join(f"{1e3 * np.median(s[:, k]):7.1f}" for k in range(7)) + f" | {1e3 * s[:, 6].mean():7.1f}")
