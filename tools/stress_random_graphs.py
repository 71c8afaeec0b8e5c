#!/usr/bin/env python3
"""randomised sweep of the batch path against the oracle: pose graphs of 300-3 600 poses with 0.3-1.6 random loop closures per pose (dense root
fronts of every width modulo the 32- and 128-column blockings), default options / every front on the multi-workgroup path / without the wide
back substitution.  Found the out-of-bounds staging reads of round 3's k_block_solve and k_backsolve_blk.  python tools/stress_random_graphs.py"""
import os, sys, numpy as np
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
base = int(sys.argv[1]) if len(sys.argv) > 1 else 7          # another argument: another set of graphs
rng = np.random.default_rng(base)
worst = 0
cases = [(int(rng.integers(300, 3600)), None, 200 * base // 7 + i) for i in range(24)]
for n, m, seed in cases:
    m = int(n * rng.uniform(0.3, 1.6))
    arr = datasets.random_pose_graph(n, m, seed)
    oc, ost = oracle.iterate(arr, 2)
    for o in (dict(), dict(small_lds_kb=0), dict(small_lds_kb=0, blk_backsolve=0)):
        for k, v in o.items(): lib.set_option(k, v)
        try:
            c, st, s = run_batch(arr, 2)
        finally:
            for k in o: lib.set_option(k, dict(small_lds_kb=156, blk_backsolve=1)[k])
        e1 = float(np.max(np.abs(c - oc) / oc)); e2 = float(np.max(np.abs(st - ost)))
        worst = max(worst, e1)
        print(f"n={n} m={m} seed={seed} {o}: rows {s['max_front_rows']} chi2 relerr {e1:.2e} states {e2:.2e}", flush=True)
        assert e1 < 1e-7 and e2 < 1e-5, "MISMATCH"
print("all ok, worst chi2 relerr", worst)
