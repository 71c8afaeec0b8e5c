#!/usr/bin/env python3
"""Where the time of the pose-by-pose demo goes: the slowest steps and the split by kind of step.  python tools/inc_slowest.py [n_poses]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
lib = host.SolverLib()
res = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=n, deterministic=True)
ms, wb = res["ms"], res["was_batch"]
order = np.argsort(-ms)
print(f"total {ms.sum():.1f} ms over {n} steps; median {np.median(ms):.4f}  mean {ms.mean():.4f}")
print("slowest steps:", [(int(i), round(float(ms[i]), 2), bool(wb[i])) for i in order[:12]])
print(f"steps with a batch fall-back: {int(wb.sum())}, {ms[wb].sum():.1f} ms (median {np.median(ms[wb]):.3f}); the others: {ms[~wb].sum():.1f} ms (median {np.median(ms[~wb]):.4f}, mean {ms[~wb].mean():.4f})")
for lo, hi in ((0, 0.05), (0.05, 0.1), (0.1, 0.3), (0.3, 1), (1, 5), (5, 1e9)):
    sel = (~wb) & (ms >= lo) & (ms < hi)
    print(f"  incremental steps of {lo}..{hi} ms: {int(sel.sum()):5d} steps, {ms[sel].sum():8.1f} ms")
