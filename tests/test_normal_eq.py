"""The normal-equation residual checker (tests/support/normal_eq.py) pinned on the CPU: against the oracle's dense normal
equations, against the oracle's own solutions, and -- when oracle/_ref is built -- against the UNMODIFIED reference's
l_point / delta_X after april_graph_cholesky.  The GPU tests (tests/test_gpu_normal_eq.py) then use it at sizes no CPU path reaches."""
import numpy as np
import pytest

from aprilsam_amd import datasets
from tests.support.normal_eq import linearise, normal_equation_residual

LAM = 1e-4          # april_graph_cholesky_param_init: tikhanov (aprilsam.c:62-75)


def _sym(A):
    return np.triu(A) + np.triu(A, 1).T


def test_residual_equals_dense_normal_equations_of_the_oracle(oracle):
    orc = oracle
    arr = datasets.random_pose_graph(60, 50, 11)
    arr = datasets.with_prior(*arr)
    s, fa, fb, z, W = arr
    A, B = orc.normal_equations(s, fa, fb, z, W, LAM)
    A = _sym(A)                                     # the oracle (like the reference) keeps the upper triangle
    rng = np.random.default_rng(5)
    dx = rng.normal(size=s.shape)
    want = (A @ dx.reshape(-1) - B).reshape(-1, 3)
    # the checker's residual, element by element
    Ja, Jb, r = linearise(s, fa, fb, z)
    got = np.zeros_like(s)
    Wm = np.asarray(W).reshape(-1, 3, 3)
    for f in range(len(fa)):
        e = Ja[f] @ dx[fa[f]] - r[f]
        if fb[f] >= 0:
            e = e + Jb[f] @ dx[fb[f]]
        got[fa[f]] += Ja[f].T @ (Wm[f] @ e)
        if fb[f] >= 0:
            got[fb[f]] += Jb[f].T @ (Wm[f] @ e)
    got += LAM * dx
    assert np.max(np.abs(got - want)) <= 1e-10 * np.max(np.abs(want))
    out = normal_equation_residual(s, fa, fb, z, W, dx, LAM)
    assert abs(out["max_abs_res"] - np.max(np.abs(want))) <= 1e-10 * np.max(np.abs(want))
    assert abs(out["max_abs_rhs"] - np.max(np.abs(B))) <= 1e-12 * np.max(np.abs(B))


@pytest.mark.parametrize("case", ["random", "lattice", "m3500"])
def test_oracle_solutions_leave_a_rounding_level_residual(case, oracle, lib):
    orc = oracle
    if case == "random":
        arr = datasets.with_prior(*datasets.random_pose_graph(400, 300, 3))
    elif case == "lattice":
        arr = lib.lattice_arrays(24)
    else:
        arr = datasets.m3500_batch()
    s, fa, fb, z, W = arr
    _, dx, _ = orc.batch_step(s, fa, fb, z, W, LAM)
    out = normal_equation_residual(s, fa, fb, z, W, dx, LAM)
    assert out["rel_max"] < 1e-9, out
    # negative control: one component of one pose off by 1e-6 is seen
    bad = dx.copy(); bad[len(bad) // 2, 0] += 1e-6
    assert normal_equation_residual(s, fa, fb, z, W, bad, LAM)["rel_max"] > 100 * max(out["rel_max"], 1e-12)
    # ... and so is a solution of the system without its Tikhonov term on a graph where it matters little: lam itself
    assert normal_equation_residual(s, fa, fb, z, W, dx, 0.0)["max_abs_res"] >= 0.5 * LAM * np.max(np.abs(dx)) - out["max_abs_res"]


@pytest.mark.parametrize("case", ["m3500", "lattice"])
def test_the_unmodified_reference_satisfies_the_checker(case, reflib, lib):
    """l_point and delta_X the reference leaves in the node objects after april_graph_cholesky (aprilsam.c:131-135, 311-315)"""
    ref = reflib
    arr = datasets.m3500_batch() if case == "m3500" else lib.lattice_arrays(32)
    g = ref.new_graph(); g.build_from_arrays(*arr); p = ref.new_param()
    for it in range(2):
        g.cholesky(p)
        lp, dx, st = g.l_points(), g.deltas(), g.states()
        out = normal_equation_residual(lp, arr[1], arr[2], arr[3], arr[4], dx, LAM)
        assert out["rel_max"] < 1e-9, (it, out)
        # state = l_point + dx, theta wrapped (april_graph_xyt.c:302-314)
        d = st - lp - dx
        d[:, 2] = (d[:, 2] + np.pi) % (2 * np.pi) - np.pi
        assert np.max(np.abs(d)) < 1e-12
    p.destroy(); g.destroy()
