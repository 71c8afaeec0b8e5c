#!/usr/bin/env python3
"""Soak of the batch API path: a mix of structured / random graphs solved over and over in random order (fresh graph + param each
time, two Gauss-Newton iterations), every result compared BITWISE with the first result of the same case (the path is deterministic:
fixed summation order everywhere).  A run that differs is re-run at once to tell a transient from a persistent difference.
    python tools/soak_batch.py [seconds] [seed] [focus]       (focus = 1: half of the runs are the two chain-like cases that showed differences)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, host
from tests.support import sweeps

T = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = host.SolverLib()
kinds = [("chain", 1500), ("chain", 97), ("star", 700), ("star", 130), ("complete", 60), ("complete", 140), ("two", 900), ("band", 1100), ("band", 257), ("comb", 1300)]
cases = [(f"{k} n={n}", sweeps.structured(k, n, 300 + i)) for i, (k, n) in enumerate(kinds)]
cases += [(f"random {i}", datasets.random_pose_graph(int(n), int(m), 50 + i)) for i, (n, m) in enumerate([(300, 400), (900, 800), (1500, 900), (2500, 2000)])]
focus = len(sys.argv) > 3 and int(sys.argv[3]) != 0
optsets = [dict()] * 8 + [dict(small_lds_kb=0), dict(small_lds_kb=16, leaf_nodes=6)] if focus else [dict(), dict(), dict(), dict(small_lds_kb=0), dict(small_lds_kb=16, leaf_nodes=6)]
first = {}; runs = 0; bad = 0; t0 = time.time(); t_rep = t0; prev = None


def run_case(arr):
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    chi2 = [g.chi2()]; flags = []
    for _ in range(2):
        g.cholesky(p); chi2.append(g.chi2()); s = p.stats(); flags.append((s["not_spd"], s["symbolic_reused"], s["reserved1"]))
    st = g.states(); p.destroy(); g.destroy()
    s["calls"] = flags
    return np.array(chi2), st, s


while time.time() - t0 < T:
    ci = int(rng.integers(len(cases))); oi = int(rng.integers(len(optsets)))
    if focus and rng.random() < 0.5: ci = 0 if rng.random() < 0.5 else 7
    label, arr = cases[ci]; o = optsets[oi]
    with lib.options(**o):
        c, st, s = run_case(arr)
    key = (ci, tuple(sorted(o.items())))
    was, prev = prev, (label, o)
    runs += 1
    if key not in first: first[key] = (c, st); continue
    if not (np.array_equal(c, first[key][0]) and np.array_equal(st, first[key][1])) or s["error_code"]:
        bad += 1
        e2 = float(np.max(np.abs(st - first[key][1])))
        with lib.options(**o):
            c2, st2, s2 = run_case(arr)
        again = np.array_equal(c2, first[key][0]) and np.array_equal(st2, first[key][1])
        print(f"run {runs} t={time.time() - t0:.1f}s DIFFERENT: {label} {o}: chi2 {c.tolist()} expected {first[key][0].tolist()} max state diff {e2:.3e} "
              f"error_code {s['error_code']} (not_spd, reused, ran twice) per call {s['calls']} fronts {s['n_fronts']} levels {s['n_levels']} previous run {was}; re-run at once {'matches again' if again else 'STILL different: ' + str(c2.tolist())}", flush=True)
    if time.time() - t_rep > 30: t_rep = time.time(); print(f"  {runs} runs, {bad} different, {time.time() - t0:.0f} s", flush=True)
print(f"soak: {runs} runs in {time.time() - t0:.0f} s, {bad} different")
