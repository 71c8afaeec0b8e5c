#!/usr/bin/env python3
"""randomised / structured sweep, not part of the test suite (minutes of GPU time): 22 random growth sequences (260 / 200 incremental steps each, several fall-back thresholds and tail sizes) against the live reference, step by step.  python tools/stress_incremental.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
from tests.support.oracle_binding import REFLIB
import tests.test_gpu_parity as T
lib = host.SolverLib(); reflib = host.SolverLib(REFLIB)
worst_c = worst_s = 0.0
for seed in range(100, 112):
    nth = [10 ** 6, 40, 12, 25][seed % 4]; steps = 260
    ours = T._random_growth(lib, seed, steps, nth)
    ref = T._random_growth(reflib, seed, steps, nth)
    ec = max(abs(a[0] - b[0]) / max(b[0], 1.0) for a, b in zip(ours, ref)); es = max(float(np.max(np.abs(a[1] - b[1]))) for a, b in zip(ours, ref))
    worst_c = max(worst_c, ec); worst_s = max(worst_s, es)
    print(f"random growth seed {seed} nthreshold {nth}: chi2 {ec:.2e} states {es:.2e}", flush=True)
    assert ec < 1e-6 and es < 1e-6
for seed, tp in zip(range(200, 210), [28, 9, 16, 12, 28, 20, 8, 28, 11, 24]):
    nth = [10 ** 6, 30][seed % 2]
    ours = T._recent_pose_growth(lib, seed, 200, nth, tp)
    ref = T._recent_pose_growth_ref(reflib, seed, 200, nth)
    ec = max(abs(a[0] - b[0]) / max(b[0], 1.0) for a, b in zip(ours, ref)); es = max(float(np.max(np.abs(a[1] - b[1]))) for a, b in zip(ours, ref))
    worst_c = max(worst_c, ec); worst_s = max(worst_s, es)
    print(f"recent-pose growth seed {seed} tail_poses {tp} nthreshold {nth}: chi2 {ec:.2e} states {es:.2e}", flush=True)
    assert ec < 1e-6 and es < 1e-6
print("all ok", worst_c, worst_s)
