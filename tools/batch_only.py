#!/usr/bin/env python3
"""The demo's --batch_update_only mode (one april_graph_cholesky per new pose): python tools/batch_only.py [n_poses]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
lib = host.SolverLib()
r = harness.run_demo(lib, datasets.m3500_arrays(), batch_update_only=True, max_poses=n)
ms = r["ms"][1:]
print(f"batch_update_only, {n} poses: total {ms.sum():.1f} ms  mean {ms.mean():.4f}  median {np.median(ms):.4f}  p99 {np.percentile(ms, 99):.3f}  max {ms.max():.2f}; final chi2 {r['chi2'][-1]:.9f}")
