#!/usr/bin/env python3
"""Per-step view of the pose-by-pose M3500 demo: time against the number of fronts a step regenerated / updated.
python tools/inc_steps.py [n_poses]      (options through APRILSAM_AMD_* as usual)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
lib = host.SolverLib()
regen = np.full(n, -1); upd = np.zeros(n, int)


def on_step(k, p, was_batch):
    if k == 0 or was_batch:
        return
    st = p.stats()
    if st["symbolic_reused"]:
        regen[k] = st["reserved0"]; upd[k] = st["inc_fronts_updated"]


res = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=n, deterministic=True, on_step=on_step)
ms, wb = res["ms"], res["was_batch"]
print(f"total {ms.sum():.1f} ms (without the first call {ms[1:].sum():.1f}); median {np.median(ms):.4f}; fall-backs {int(wb.sum()) - 1}: {ms[wb][1:].sum():.1f} ms")
inc = (~wb) & (regen >= 0)
print(f"incremental steps on the frozen plan: {int(inc.sum())}, {ms[inc].sum():.1f} ms; re-planned: {int(((~wb) & (regen < 0)).sum())} steps, {ms[(~wb) & (regen < 0)].sum():.1f} ms")
print("fronts regenerated -> steps, total ms, median ms | of them with updated fronts: steps, ms, median, mean updated")
for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 7), (7, 10), (10, 14), (14, 20), (20, 1000)):
    sel = inc & (regen >= lo) & (regen < hi)
    if not sel.any():
        continue
    su = sel & (upd > 0)
    line = f"  {lo:3d}..{hi - 1:<4d} {int(sel.sum()):5d} steps {ms[sel].sum():8.1f} ms  median {np.median(ms[sel]):.4f}"
    if su.any():
        line += f" | {int(su.sum()):5d} steps {ms[su].sum():8.1f} ms  median {np.median(ms[su]):.4f}  updated {upd[su].mean():.1f} of {regen[su].mean():.1f}"
    print(line)
