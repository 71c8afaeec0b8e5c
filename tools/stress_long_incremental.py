#!/usr/bin/env python3
"""randomised sweep, not part of the test suite: 1 200 incremental steps (1 372 poses, dense loop closures) with and without fall-backs against the live reference -- the REFERENCE needs 3 minutes per run without fall-backs (ours: 0.6 s): 7 minutes of box time.  python tools/stress_long_incremental.py"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
from tests.support.oracle_binding import REFLIB
import tests.test_gpu_parity as T
lib = host.SolverLib(); reflib = host.SolverLib(REFLIB)
for seed, nth in ((400, 10 ** 6), (401, 150), (402, 10 ** 6), (403, 60)):
    t0 = time.time(); ours = T._random_growth(lib, seed, 1200, nth, observe_every=10); t1 = time.time()
    ref = T._random_growth(reflib, seed, 1200, nth, observe_every=10); t2 = time.time()
    ec = max(abs(a[0] - b[0]) / max(b[0], 1.0) for a, b in zip(ours, ref)); es = max(float(np.max(np.abs(a[1] - b[1]))) for a, b in zip(ours, ref))
    print(f"long growth seed {seed} nthreshold {nth}: {len(ours[-1][1])} poses, chi2 {ec:.2e} states {es:.2e}  (ours {t1 - t0:.1f} s incl. python, reference {t2 - t1:.1f} s)", flush=True)
    assert ec < 1e-6 and es < 1e-6
print("all ok")
