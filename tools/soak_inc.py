#!/usr/bin/env python3
"""Soak of the incremental path's multi-level launches (regenerated fronts + low-rank updates + partial back substitution, dependency flags that
carry the step number): the first n poses of the M3500 demo over and over under rotating option sets -- the default (most small steps in one
workgroup), inc_one = 0 (EVERY step through the flag-synchronised launches), the same with pool_poison = 1 -- every chi^2 trace, fall-back
schedule and final state compared BITWISE with the first run of its option set, and the first runs with the reference golden.
    python tools/soak_inc.py [seconds] [ignored: tools/soak_matrix.sh passes a seed]          (SOAK_INC_POSES=n: poses per run, default 700)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host
T = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n = int(os.environ.get("SOAK_INC_POSES", "700"))
lib = host.SolverLib()
G = np.load(os.path.join(ROOT, "tests", "golden", "m3500_inc_demo.npz"))
arr = datasets.m3500_arrays()
OPTS = [dict(), dict(inc_one=0, inc_tail=0), dict(inc_one=0, inc_tail=0, pool_poison=1), dict(inc_update=0, inc_one=0), dict(pool_poison=1)]
first = {}; runs = steps = bad = 0; t0 = time.time(); k = 0
while time.time() - t0 < T:
    o = OPTS[k % len(OPTS)]; k += 1
    with lib.options(**o):
        r = harness.run_demo(lib, arr, max_poses=n, deterministic=True)
    runs += 1; steps += n
    key = tuple(sorted(o.items()))
    if key not in first:
        rel = float(np.max(np.abs(r["chi2"] - G["chi2"][:n]) / np.maximum(G["chi2"][:n], 1e-9)))
        ok = rel < 1e-6 and np.array_equal(r["was_batch"], G["was_batch"][:n]) and np.all(np.isfinite(r["final_states"]))
        print(f"first run {o}: chi2 max rel err vs the reference golden {rel:.2e}, schedule identical {bool(np.array_equal(r['was_batch'], G['was_batch'][:n]))}", flush=True)
        bad += 0 if ok else 1
        first[key] = r; continue
    f = first[key]
    if not (np.array_equal(r["chi2"], f["chi2"]) and np.array_equal(r["was_batch"], f["was_batch"]) and np.array_equal(r["final_states"], f["final_states"])):
        bad += 1
        d = np.nonzero(r["chi2"] != f["chi2"])[0]
        print(f"run {runs} DIFFERENT {o}: first differing step {int(d[0]) if len(d) else -1}, max |state diff| {float(np.max(np.abs(r['final_states'] - f['final_states']))):.3e}", flush=True)
print(f"soak_inc: {runs} runs of {n} poses = {steps} incremental steps in {time.time() - t0:.0f} s, {bad} different")
