"""Host logic (no GPU): nested-dissection ordering + symbolic plan of the product library, checked
structurally and numerically (numpy emulation of the device algorithm vs the oracle's sparse solve)."""
import numpy as np
import pytest

from aprilsam_amd import datasets
from tests.support.mf_emulator import PlanView, contributions, solve


def _check_structure(P, N, fa, fb):
    assert sorted(P.perm.tolist()) == list(range(N))
    assert np.array_equal(P.pos[P.perm], np.arange(N))
    nF = P.nF
    # own blocks tile the positions in front order
    assert P.front_first[0] == 0
    assert np.array_equal(P.front_first[1:], np.cumsum(P.front_nsb)[:-1])
    assert int(P.front_nsb.sum()) == N
    for t in range(nF):
        rows = P.front_rows[P.front_rows_ptr[t]:P.front_rows_ptr[t + 1]]
        last = P.front_first[t] + P.front_nsb[t] - 1
        assert np.all(np.diff(rows) > 0) and (len(rows) == 0 or rows[0] > last)
        par = P.front_parent[t]
        if len(rows):
            assert par > t and P.front_level[par] > P.front_level[t]
            # parent contains every struct row of the child (own or struct): rel indexes it
            prow = np.concatenate([np.arange(P.front_first[par], P.front_first[par] + P.front_nsb[par]),
                                   P.front_rows[P.front_rows_ptr[par]:P.front_rows_ptr[par + 1]]])
            rel = P.front_rel[P.front_rows_ptr[t]:P.front_rows_ptr[t + 1]]
            assert np.array_equal(prow[rel], rows)
        else:
            assert par == -1
    # every factor sits inside its owner front, owner = front of its earliest-eliminated node
    for f in range(len(fa)):
        t = P.factor_front[f]
        pa = P.pos[fa[f]]; pb = P.pos[fb[f]] if fb[f] >= 0 else pa
        rows = np.concatenate([np.arange(P.front_first[t], P.front_first[t] + P.front_nsb[t]),
                               P.front_rows[P.front_rows_ptr[t]:P.front_rows_ptr[t + 1]]])
        assert rows[P.factor_la[f]] == pa
        if fb[f] >= 0:
            assert rows[P.factor_lb[f]] == pb
        assert P.front_first[t] <= min(pa, pb) < P.front_first[t] + P.front_nsb[t]
    # levels partition the fronts
    assert sorted(P.lev_fronts.tolist()) == list(range(nF))


@pytest.mark.parametrize("case", ["random_small", "random_mid", "lattice12", "m3500", "disconnected", "star"])
def test_plan_structure_and_numeric_emulation(lib, oracle, case):
    leaf = 16
    if case == "random_small":
        arr = datasets.random_pose_graph(40, 30, 5); leaf = 4
    elif case == "random_mid":
        arr = datasets.random_pose_graph(600, 500, 6); leaf = 8
    elif case == "lattice12":
        arr = lib.lattice_arrays(12); leaf = 6
    elif case == "m3500":
        arr = datasets.m3500_batch()
    elif case == "disconnected":    # two components + an isolated node, each anchored by its own prior
        a = datasets.random_pose_graph(30, 10, 1); b = datasets.random_pose_graph(25, 8, 2)
        st = np.vstack([a[0], b[0], [[1.0, 2.0, 0.5]]])
        off = len(a[0])
        fb_b = np.where(b[2] >= 0, b[2] + off, -1)
        arr = (st, np.concatenate([a[1], b[1] + off]).astype(np.int32), np.concatenate([a[2], fb_b]).astype(np.int32),
               np.vstack([a[3], b[3]]), np.vstack([a[4], b[4]]))
        leaf = 5
    else:                            # star: node 0 connected to everybody (separator = hub)
        rng = np.random.default_rng(0); n = 60
        st = np.column_stack([rng.normal(size=n), rng.normal(size=n), rng.uniform(-3, 3, n)])
        fa = np.zeros(n - 1, np.int32); fb = np.arange(1, n, dtype=np.int32)
        z = rng.normal(size=(n - 1, 3)); W = np.tile(np.diag([10.0, 10.0, 5.0]).reshape(9), (n - 1, 1))
        arr = datasets.with_prior(st, fa, fb, z, W); leaf = 4
    st, fa, fb, z, W = arr
    N = len(st)
    P = PlanView(lib, N, fa, fb, xy=st[:, :2], leaf_nodes=leaf)
    _check_structure(P, N, fa, fb)
    lam = np.full(N, 1e-4)
    H, G = contributions(oracle, st, st, fa, fb, z, W, P.factor_swap)
    x = solve(P, H, G, lam)
    dx = x.reshape(N, 3)[P.pos]                       # node order
    ref = oracle.solve_system(st, st, fa, fb, z, W, lam)
    scale = max(1.0, np.max(np.abs(ref)))
    assert np.max(np.abs(dx - ref)) < 1e-7 * scale


def test_plan_without_coordinates_and_leaf_sizes(lib):
    st, fa, fb, z, W = lib.lattice_arrays(20)
    for leaf in (1, 3, 16, 64, 1000):
        P = PlanView(lib, len(st), fa, fb, xy=None, leaf_nodes=leaf)
        _check_structure(P, len(st), fa, fb)
        if leaf >= 400:
            assert P.nF == 1       # the whole graph is one dense front


def test_lattice_plan_quality(lib):
    """nested dissection on the 100k lattice must beat the reference ordering's fill (SURVEY.md §6:
    nnz(U) 62.9 M, sum c^2 6.45e10) — this is where the GPU path gets its flop advantage from."""
    st, fa, fb, z, W = lib.lattice_arrays(316)
    P = PlanView(lib, len(st), fa, fb, xy=st[:, :2], leaf_nodes=16)
    nnzL, flops = int(P.stats[3]), float(P.stats[4])
    assert nnzL < 62.9e6 and flops < 3.0e10
    assert P.nLevels <= 16


@pytest.mark.parametrize("case", ["m3500", "random_mid", "lattice40"])
def test_separator_amalgamation_keeps_the_plan_valid(lib, oracle, case):
    """option amalg (symbolic.cpp amalgamate: separator fronts take in child separators where a model of the critical path says so --
    measured on the box and left off, profiles/r06_experiments_not_kept.txt): fewer fronts and levels, still a valid plan, and the numeric
    emulation still solves the system"""
    arr = {"m3500": datasets.m3500_batch, "random_mid": lambda: datasets.random_pose_graph(600, 500, 6), "lattice40": lambda: lib.lattice_arrays(40)}[case]()
    st, fa, fb, z, W = arr
    N = len(st)
    P0 = PlanView(lib, N, fa, fb, xy=st[:, :2], leaf_nodes=16)
    with lib.options(amalg=1, amalg_max=48):
        P = PlanView(lib, N, fa, fb, xy=st[:, :2], leaf_nodes=16)
    _check_structure(P, N, fa, fb)
    assert P.nF <= P0.nF and P.nLevels <= P0.nLevels
    if case == "m3500":
        assert P.nLevels < P0.nLevels and int(P.front_nsb.max()) <= 48
    H, G = contributions(oracle, st, st, fa, fb, z, W, P.factor_swap)
    lam = np.full(N, 1e-4)
    dx = solve(P, H, G, lam).reshape(N, 3)[P.pos]
    ref = oracle.solve_system(st, st, fa, fb, z, W, lam)
    assert np.max(np.abs(dx - ref)) < 1e-7 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("pin", [1, 8, 40])
def test_pin_last_keeps_the_newest_poses_in_the_root_front(lib, oracle, pin):
    """option pin_last ("recent poses last", cf. aprilsam.c:1021-1098): the k newest poses form the root front; the plan
    stays a valid plan and the numeric emulation still solves the system"""
    st, fa, fb, z, W = datasets.random_pose_graph(200, 120, 9)
    N = len(st)
    lib.set_option("pin_last", pin)
    try:
        P = PlanView(lib, N, fa, fb, xy=st[:, :2], leaf_nodes=8)
    finally:
        lib.set_option("pin_last", 0)
    _check_structure(P, N, fa, fb)
    root = P.nF - 1
    assert P.front_parent[root] == -1 and P.front_nsb[root] == pin
    assert sorted(P.perm[P.front_first[root]:].tolist()) == list(range(N - pin, N))
    H, G = contributions(oracle, st, st, fa, fb, z, W, P.factor_swap)
    lam = np.full(N, 1e-4)
    dx = solve(P, H, G, lam).reshape(N, 3)[P.pos]
    ref = oracle.solve_system(st, st, fa, fb, z, W, lam)
    assert np.max(np.abs(dx - ref)) < 1e-7 * max(1.0, np.max(np.abs(ref)))


_PLAN_HASH_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, {root!r})
from aprilsam_amd import datasets, host
from tests.support.mf_emulator import PlanView
lib = host.SolverLib()
out = []
cases = [datasets.m3500_batch(), lib.lattice_arrays(60), datasets.random_pose_graph(1500, 700, 5)]
for s, fa, fb, z, W in cases:
    P = PlanView(lib, len(s), fa, fb, xy=s[:, :2], leaf_nodes=16)
    h = hashlib.sha256()
    for name in ("perm", "front_first", "front_nsb", "front_parent", "front_rows", "lev_fronts"):
        h.update(np.ascontiguousarray(getattr(P, name)).tobytes())
    out.append(h.hexdigest()[:16])
print(" ".join(out))
"""


def test_plan_does_not_depend_on_the_number_of_planner_threads():
    """ordering.cpp: the top of the dissection tree is computed level by level on a pool of threads (regions side by side,
    candidate splits side by side) and numbered afterwards in the serial order; the subtrees below are private per task.
    The plan must be bit-identical for every thread count (all ranks of a sharded run build it independently).  One process
    per count: the pool is sized when it is first used."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    for nt in ("1", "2", "5", "16"):
        env = dict(os.environ, APRILSAM_AMD_PLAN_THREADS=nt)
        r = subprocess.run([sys.executable, "-c", _PLAN_HASH_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        seen[nt] = r.stdout.strip().splitlines()[-1]
    assert len(set(seen.values())) == 1, seen
