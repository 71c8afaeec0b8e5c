#!/usr/bin/env python3
"""where a fall-back step of the incremental demo spends its time (stats of the batch call inside): python tools/fallback_phases.py [n_poses]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host
lib = host.SolverLib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
rows = []
def cb(k, p, was_batch):
    if was_batch:
        st = p.stats()
        rows.append((k, 0.0, st["ms_pack"], st["ms_symbolic"], st["ms_h2d"], st["ms_device"], st["ms_unpack"], st["ms_total"], st["n_nodes"]))
res = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=n, deterministic=True, on_step=cb)
for r in rows: pass
rows = [(r[0], res['ms'][r[0]]) + r[2:] for r in rows]
a = np.array(rows)
print("fall-back steps:", len(a))
print("step  wall  pack  symbolic  h2d  device  unpack  total(inside)  nodes")
for r in a[:: max(1, len(a) // 12)]:
    print("%5d %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f %6d" % tuple(r))
print("sum  wall %.1f pack %.1f symbolic %.1f h2d %.1f device %.1f unpack %.1f inside %.1f ms" % tuple(a[:, 1:8].sum(0)))
