#!/usr/bin/env python3
"""randomised / structured sweep, not part of the test suite (minutes of GPU time): lattices of odd sizes (K = 37 .. 170) against the oracle, four option sets.  python tools/stress_lattices.py"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, host
from tests.support.oracle_binding import Oracle
lib = host.SolverLib(); oracle = Oracle()
def run_batch(arr, iters):
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    chi2 = [g.chi2()]
    for _ in range(iters):
        g.cholesky(p); chi2.append(g.chi2())
    st = g.states(); s = p.stats(); p.destroy(); g.destroy()
    return np.array(chi2), st, s
for K in (37, 91, 131, 170):
    arr = lib.lattice_arrays(K)
    t0 = time.time(); oc, ost = oracle.iterate(arr, 2); to = time.time() - t0
    for o in (dict(), dict(small_lds_kb=0), dict(small_lds_kb=48, leaf_nodes=24), dict(small_lds_kb=0, leaf_nodes=7, syrk_xcd_order=1, syrk_small_tiles=1 << 30)):
        for k, v in o.items(): lib.set_option(k, v)
        try:
            c, st, s = run_batch(arr, 2)
        finally:
            for k in o: lib.set_option(k, dict(small_lds_kb=156, leaf_nodes=16, syrk_xcd_order=512, syrk_small_tiles=320)[k])
        e1 = float(np.max(np.abs(c - oc) / oc)); e2 = float(np.max(np.abs(st - ost)))
        print(f"lattice K={K} (oracle {to:.1f} s) {o}: fronts {s['n_fronts']} levels {s['n_levels']} rows {s['max_front_rows']} chi2 relerr {e1:.2e} states {e2:.2e}", flush=True)
        assert e1 < 1e-8 and e2 < 1e-6, "MISMATCH"
print("all ok")
