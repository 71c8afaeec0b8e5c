"""Randomised / structured sweeps of the HIP path against the oracle and the live reference.  TEST INFRASTRUCTURE.
tools/stress_*.py run them at full size by hand (minutes of GPU time each; they found round 4's out-of-bounds staging reads);
tests/test_gpu_sweeps.py runs a bounded slice of every one under `-m gpu`, one of them with the pool_guard option (NaN-filled
guard bands behind every frontal array: a stray read that is used changes the result, a stray write is reported)."""
import numpy as np

from aprilsam_amd import datasets


def run_batch(lib, arr, iters):
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    chi2 = [g.chi2()]
    for _ in range(iters):
        g.cholesky(p); chi2.append(g.chi2())
    st = g.states(); s = p.stats(); p.destroy(); g.destroy()
    return np.array(chi2), st, s


def structured(kind, n, seed):
    """chains, stars, complete graphs, two components, banded and comb graphs with full information matrices"""
    rng = np.random.default_rng(seed)
    st = np.column_stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-np.pi, np.pi, n)])
    if kind == "chain": pairs = [(i, i + 1) for i in range(n - 1)]
    elif kind == "star": pairs = [(0, i) for i in range(1, n)]
    elif kind == "complete": pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    elif kind == "two": h = n // 2; pairs = [(i, i + 1) for i in range(h - 1)] + [(i, i + 1) for i in range(h, n - 1)] + [(0, h - 1), (h, n - 1)]
    elif kind == "band": pairs = [(i, i + d) for d in (1, 2, 3, 7) for i in range(n - d)]
    elif kind == "comb": pairs = [(i, i + 1) for i in range(n - 1)] + [(i, (i * 37) % n) for i in range(0, n, 3) if i != (i * 37) % n]
    else: raise ValueError(kind)
    pairs = sorted(set(tuple(sorted(p)) for p in pairs if p[0] != p[1]))
    fa = np.array([p[0] for p in pairs], np.int32); fb = np.array([p[1] for p in pairs], np.int32)
    F = len(pairs); z = np.empty((F, 3)); W = np.empty((F, 9))
    for k in range(F):
        pa, pb = st[fa[k]], st[fb[k]]; c, s = np.cos(pa[2]), np.sin(pa[2]); dx, dy = pb[0] - pa[0], pb[1] - pa[1]
        z[k] = [c * dx + s * dy + rng.normal(0, .3), -s * dx + c * dy + rng.normal(0, .3), pb[2] - pa[2] + rng.normal(0, .1)]
        M = rng.normal(size=(3, 3)); Wk = M @ M.T + np.diag([20., 20., 50.]); W[k] = ((Wk + Wk.T) / 2).reshape(9)
    if kind == "two":     # a prior on each component
        s2, a2, b2, z2, W2 = datasets.with_prior(st, fa, fb, z, W, first=True)
        a2 = np.append(a2, n // 2).astype(np.int32); b2 = np.append(b2, -1).astype(np.int32); z2 = np.vstack([z2, st[n // 2]]); W2 = np.vstack([W2, W2[0]])
        return s2, a2, b2, z2, W2
    return datasets.with_prior(st, fa, fb, z, W, first=True)


def sweep_batch(lib, oracle, cases, option_sets, chi2_tol, state_tol, log=print):
    """cases: iterable of (label, arrays); every option set on every case against two oracle iterations"""
    worst = 0.0
    for label, arr in cases:
        oc, ost = oracle.iterate(arr, 2)
        for o in option_sets:
            with lib.options(**o):
                c, st, s = run_batch(lib, arr, 2)
            e1 = float(np.max(np.abs(c - oc) / np.maximum(oc, 1e-12))); e2 = float(np.max(np.abs(st - ost)))
            worst = max(worst, e1)
            log(f"{label} {o}: fronts {s['n_fronts']} levels {s['n_levels']} rows {s['max_front_rows']} chi2 relerr {e1:.2e} states {e2:.2e}")
            assert s["error_code"] == 0 and s["not_spd"] == 0, (label, o, s)
            assert e1 < chi2_tol and e2 < state_tol, ("MISMATCH", label, o, e1, e2, c.tolist(), oc.tolist(), s)
    return worst


def random_graph_cases(base, count, n_lo=300, n_hi=3600):
    rng = np.random.default_rng(base)
    out = []
    for i in range(count):
        n = int(rng.integers(n_lo, n_hi)); seed = 200 * base // 7 + i; m = int(n * rng.uniform(0.3, 1.6))
        out.append((f"random n={n} m={m} seed={seed}", datasets.random_pose_graph(n, m, seed)))
    return out


def compare_traces(ours, ref):
    ec = max(abs(a[0] - b[0]) / max(b[0], 1.0) for a, b in zip(ours, ref))
    es = max(float(np.max(np.abs(a[1] - b[1]))) for a, b in zip(ours, ref))
    return ec, es


# ---- soak of the batch API path (tools/soak_batch.py; a bounded slice runs under -m gpu) --------------------------------------
SOAK_KINDS = [("chain", 1500), ("chain", 97), ("star", 700), ("star", 130), ("complete", 60), ("complete", 140), ("two", 900), ("band", 1100), ("band", 257), ("comb", 1300)]
SOAK_OPTIONS = [dict(), dict(), dict(), dict(small_lds_kb=0), dict(small_lds_kb=16, leaf_nodes=6)]


def soak_cases():
    cases = [(f"{k} n={n}", structured(k, n, 300 + i)) for i, (k, n) in enumerate(SOAK_KINDS)]
    return cases + [(f"random {i}", datasets.random_pose_graph(int(n), int(m), 50 + i)) for i, (n, m) in enumerate([(300, 400), (900, 800), (1500, 900), (2500, 2000)])]


def soak_run(lib, arr):
    """two april_graph_cholesky calls on a fresh graph + param: chi^2 trace, final states, last stats + (not_spd, reused, ran twice) per call"""
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    chi2 = [g.chi2()]; flags = []
    for _ in range(2):
        g.cholesky(p); chi2.append(g.chi2()); s = p.stats(); flags.append((s["not_spd"], s["symbolic_reused"], s["reserved1"]))
    st = g.states(); p.destroy(); g.destroy()
    s["calls"] = flags
    return np.array(chi2), st, s


def soak(lib, seconds, seed, log=print, cases=None, focus=False):
    """cases in random order until the time is up, every result compared BITWISE with the first of its (case, options); a run that
    differs is repeated at once (transient or persistent?).  Returns (runs, differing runs)."""
    import time
    rng = np.random.default_rng(seed)
    cases = cases or soak_cases()
    optsets = [dict()] * 8 + SOAK_OPTIONS[3:] if focus else SOAK_OPTIONS
    first = {}; runs = 0; bad = 0; t0 = time.time(); t_rep = t0; prev = None
    while time.time() - t0 < seconds:
        ci = int(rng.integers(len(cases))); oi = int(rng.integers(len(optsets)))
        if focus and rng.random() < 0.5: ci = 0 if rng.random() < 0.5 else 7
        label, arr = cases[ci]; o = optsets[oi]
        with lib.options(**o):
            c, st, s = soak_run(lib, arr)
        key = (ci, tuple(sorted(o.items())))
        was, prev = prev, (label, o)
        runs += 1
        if key not in first: first[key] = (c, st); continue
        if not (np.array_equal(c, first[key][0]) and np.array_equal(st, first[key][1])) or s["error_code"]:
            bad += 1
            e2 = float(np.max(np.abs(st - first[key][1])))
            with lib.options(**o):
                c2, st2, s2 = soak_run(lib, arr)
            again = np.array_equal(c2, first[key][0]) and np.array_equal(st2, first[key][1])
            log(f"run {runs} t={time.time() - t0:.1f}s DIFFERENT: {label} {o}: chi2 {c.tolist()} expected {first[key][0].tolist()} max state diff {e2:.3e} "
                f"error_code {s['error_code']} (not_spd, reused, ran twice) per call {s['calls']} fronts {s['n_fronts']} levels {s['n_levels']} previous run {was}; "
                f"re-run at once {'matches again' if again else 'STILL different: ' + str(c2.tolist())}")
        if time.time() - t_rep > 30: t_rep = time.time(); log(f"  {runs} runs, {bad} different, {time.time() - t0:.0f} s")
    return runs, bad


# ---- the same graphs under the debug option pool_poison (NaN in everything a multi-level launch hands over) -----------------------
def poison_soak(lib, rounds, calls, seed, log=print, cases=None):
    """Every soak case `rounds` times in random order, fresh graph + param each time, `calls` april_graph_cholesky calls in a row (cold,
    then warm -- the captured graph -- on states that keep changing), with option pool_poison on: before every call the update block of
    every front and x are NaN-filled, so a dependency wait that passes early yields NaN / "not positive definite" with certainty instead
    of the previous call's numbers.  Every chi^2 trace and final state is compared BITWISE with the first of its (case, options).
    Returns (solves, differing runs)."""
    rng = np.random.default_rng(seed)
    cases = cases or soak_cases()
    first = {}; solves = 0; bad = 0
    order = [(ci, oi) for ci in range(len(cases)) for oi in (0, 3, 4)] * rounds
    rng.shuffle(order)
    for ci, oi in order:
        label, arr = cases[ci]; o = dict(SOAK_OPTIONS[oi], pool_poison=1)
        with lib.options(**o):
            g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
            chi2 = [g.chi2()]; spd = 0
            for _ in range(calls):
                g.cholesky(p); s = p.stats(); spd += s["not_spd"]
                assert s["error_code"] == 0, (label, o, s)
            chi2.append(g.chi2()); st = g.states(); p.destroy(); g.destroy()
        solves += calls
        key = (ci, oi); c = np.array(chi2)
        if spd or not np.all(np.isfinite(st)) or not np.all(np.isfinite(c)):
            bad += 1; log(f"POISON SEEN: {label} {o}: not_spd {spd} chi2 {c.tolist()} finite states {bool(np.all(np.isfinite(st)))}"); continue
        if key not in first: first[key] = (c, st); continue
        if not (np.array_equal(c, first[key][0]) and np.array_equal(st, first[key][1])):
            bad += 1; log(f"DIFFERENT: {label} {o}: chi2 {c.tolist()} expected {first[key][0].tolist()} max state diff {float(np.max(np.abs(st - first[key][1]))):.3e}")
    return solves, bad
