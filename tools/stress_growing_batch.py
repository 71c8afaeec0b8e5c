#!/usr/bin/env python3
"""randomised sweep, not part of the test suite: random growth with ONE BATCH CALL per step (the demo's --batch_update_only; with and without batch_extend, with factors between two old poses) against the live reference, step by step.  python tools/stress_growing_batch.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
from tests.support.oracle_binding import REFLIB
import tests.test_gpu_parity as T
lib = host.SolverLib(); reflib = host.SolverLib(REFLIB)
G = type(lib.new_graph())
orig = G.cholesky_inc
G.cholesky_inc = lambda self, p: self.cholesky(p)          # the same growth, one BATCH call per step (the demo's --batch_update_only)
try:
    for seed in range(300, 308):
        for ext in (1, 0):
            lib.set_option("batch_extend", ext)
            ours = T._random_growth(lib, seed, 220, 10 ** 6, old_old=(seed % 2 == 0))
            ref = T._random_growth(reflib, seed, 220, 10 ** 6, old_old=(seed % 2 == 0))
            ec = max(abs(a[0] - b[0]) / max(b[0], 1.0) for a, b in zip(ours, ref)); es = max(float(np.max(np.abs(a[1] - b[1]))) for a, b in zip(ours, ref))
            print(f"batch call per step, seed {seed} batch_extend {ext}: chi2 {ec:.2e} states {es:.2e}", flush=True)
            assert ec < 1e-6 and es < 1e-6
finally:
    G.cholesky_inc = orig; lib.set_option("batch_extend", 1)
print("all ok")
