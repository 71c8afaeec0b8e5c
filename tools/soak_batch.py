#!/usr/bin/env python3
"""Soak of the batch API path: a mix of structured / random graphs solved over and over in random order (fresh graph + param each
time, two Gauss-Newton iterations), every result compared BITWISE with the first result of the same case (the path is deterministic:
fixed summation order everywhere).  A run that differs is re-run at once to tell a transient from a persistent difference.  This is
what found the early flag hand-overs of round 5 (profiles/r05_flag_soak.txt); tools/soak_matrix.sh runs several side by side.
    python tools/soak_batch.py [seconds] [seed] [focus]       (focus = 1: half of the runs are the two chain-like cases that showed differences)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
from tests.support import sweeps

T = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
focus = len(sys.argv) > 3 and int(sys.argv[3]) != 0
t0 = time.time()
runs, bad = sweeps.soak(host.SolverLib(), T, seed, log=lambda m: print(m, flush=True), focus=focus)
print(f"soak: {runs} runs in {time.time() - t0:.0f} s, {bad} different")
