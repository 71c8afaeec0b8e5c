#!/usr/bin/env python3
"""randomised / structured sweep, not part of the test suite (minutes of GPU time): chains, stars, complete graphs, two components, banded and comb graphs against the oracle (default options / every front on the multi-workgroup path / tiny LDS budget and leaves).  python tools/stress_structured_graphs.py"""
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
def structured(kind, n, seed):
    rng = np.random.default_rng(seed)
    st = np.column_stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-np.pi, np.pi, n)])
    if kind == "chain": pairs = [(i, i + 1) for i in range(n - 1)]
    elif kind == "star": pairs = [(0, i) for i in range(1, n)]
    elif kind == "complete": pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    elif kind == "two": h = n // 2; pairs = [(i, i + 1) for i in range(h - 1)] + [(i, i + 1) for i in range(h, n - 1)] + [(0, h - 1), (h, n - 1)]
    elif kind == "band": pairs = [(i, i + d) for d in (1, 2, 3, 7) for i in range(n - d)]
    elif kind == "comb": pairs = [(i, i + 1) for i in range(n - 1)] + [(i, (i * 37) % n) for i in range(0, n, 3) if i != (i * 37) % n]
    pairs = sorted(set(tuple(sorted(p)) for p in pairs if p[0] != p[1]))
    fa = np.array([p[0] for p in pairs], np.int32); fb = np.array([p[1] for p in pairs], np.int32)
    F = len(pairs); z = np.empty((F, 3)); W = np.empty((F, 9))
    for k in range(F):
        pa, pb = st[fa[k]], st[fb[k]]; c, s = np.cos(pa[2]), np.sin(pa[2]); dx, dy = pb[0] - pa[0], pb[1] - pa[1]
        z[k] = [c * dx + s * dy + rng.normal(0, .3), -s * dx + c * dy + rng.normal(0, .3), pb[2] - pa[2] + rng.normal(0, .1)]
        M = rng.normal(size=(3, 3)); Wk = M @ M.T + np.diag([20., 20., 50.]); W[k] = ((Wk + Wk.T) / 2).reshape(9)
    if kind == "two":     # a prior on each component
        arr = datasets.with_prior(st, fa, fb, z, W, first=True)
        s2, a2, b2, z2, W2 = arr
        a2 = np.append(a2, n // 2).astype(np.int32); b2 = np.append(b2, -1).astype(np.int32); z2 = np.vstack([z2, st[n // 2]]); W2 = np.vstack([W2, W2[0] if b2[0] < 0 else np.diag([1e3, 1e3, 1e3]).reshape(9)])
        return s2, a2, b2, z2, W2
    return datasets.with_prior(st, fa, fb, z, W, first=True)
worst = 0
cases = [("chain", 5000), ("chain", 97), ("star", 1500), ("star", 130), ("complete", 60), ("complete", 140), ("two", 2000), ("band", 3000), ("band", 257), ("comb", 2500), ("comb", 4100)]
for i, (kind, n) in enumerate(cases):
    arr = structured(kind, n, 300 + i)
    oc, ost = oracle.iterate(arr, 2)
    for o in (dict(), dict(small_lds_kb=0), dict(small_lds_kb=16, leaf_nodes=6)):
        for k, v in o.items(): lib.set_option(k, v)
        try:
            c, st, s = run_batch(arr, 2)
        finally:
            for k in o: lib.set_option(k, dict(small_lds_kb=156, leaf_nodes=16)[k])
        e1 = float(np.max(np.abs(c - oc) / np.maximum(oc, 1e-12))); e2 = float(np.max(np.abs(st - ost)))
        worst = max(worst, e1)
        print(f"{kind} n={n} {o}: fronts {s['n_fronts']} levels {s['n_levels']} rows {s['max_front_rows']} chi2 relerr {e1:.2e} states {e2:.2e}", flush=True)
        assert e1 < 1e-6 and e2 < 1e-5, "MISMATCH"
print("all ok, worst chi2 relerr", worst)
