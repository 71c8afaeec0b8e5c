#!/usr/bin/env python3
"""Repeat one structured case (fresh graph + param every time) and report every run whose chi2 trace differs from the first:
a hunt for results that depend on timing or on what ran before.   python tools/repeat_case.py [kind n seed reps interleave]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
from tests.support import sweeps
kind = sys.argv[1] if len(sys.argv) > 1 else "band"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 307; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
inter = int(sys.argv[5]) if len(sys.argv) > 5 else 1
lib = host.SolverLib()
arr = sweeps.structured(kind, n, seed)
other = sweeps.structured("two", 900, 306)
ref = None; bad = 0
for r in range(reps):
    if inter:
        for o in (dict(small_lds_kb=0), dict(small_lds_kb=16, leaf_nodes=6)):
            with lib.options(**o):
                sweeps.run_batch(lib, other, 2)
    c, st, s = sweeps.run_batch(lib, arr, 2)
    if ref is None: ref = (c, st); print("first:", c.tolist(), s["n_fronts"], s["n_levels"], s["max_front_rows"], flush=True)
    e1 = float(np.max(np.abs(c - ref[0]) / ref[0])); e2 = float(np.max(np.abs(st - ref[1])))
    if e1 > 1e-9 or e2 > 1e-7 or s["error_code"]:
        bad += 1; print(f"run {r}: DIFFERENT chi2 {c.tolist()} relerr {e1:.3e} states {e2:.3e} stats {s}", flush=True)
print(f"{reps} runs, {bad} different")
