#!/usr/bin/env python3
"""incremental steps that were handed to a full re-plan (a front outgrew the LDS, ...): what they cost: python tools/replan_phases.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host
lib = host.SolverLib()
rows = []
def cb(k, p, was_batch):
    st = p.stats()
    if not was_batch and st.get("inc_replanned"):
        rows.append((k, st["ms_pack"], st["ms_symbolic"], st["ms_h2d"], st["ms_device"], st["ms_unpack"], st["ms_total"], st["n_nodes"]))
res = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=3500, deterministic=True, on_step=cb)
print("re-planned incremental steps:", len(rows), " wall ms:", " ".join(f"{res['ms'][r[0]]:.2f}" for r in rows))
print("step  pack  symbolic  h2d  device  unpack  total(inside)  nodes")
for r in rows:
    print("%5d %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f %6d" % r)
