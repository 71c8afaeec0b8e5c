"""Pins the CPU oracle (oracle/oracle.c) against golden vectors produced by the unmodified reference
(oracle/gen_golden.py).  Runs without a GPU."""
import numpy as np
import pytest

from aprilsam_amd import datasets
from tests.conftest import golden


def test_mod2pi_range_and_edges(oracle):
    for v in [0.0, 3.0, -3.0, np.pi, -np.pi, 3 * np.pi, 7.0, -7.0, 100.0]:
        w = oracle.mod2pi(v)
        assert -np.pi <= w < np.pi
        assert abs(np.sin(w) - np.sin(v)) < 1e-12 and abs(np.cos(w) - np.cos(v)) < 1e-12
    assert oracle.mod2pi(np.pi) == -np.pi        # [-pi, pi): math_util.h:119-122


def test_factor_eval_matches_reference(oracle):
    G = golden("factor_eval.npz")
    n = len(G["pa"])
    for f in range(2 * n):
        k = f % n
        J0, J1, r, c = oracle.factor_eval(G["pa"][k], G["pb"][k] if f < n else None, G["z"][k], G["W"][k])
        assert np.array_equal(J0, G["J0"][f])
        if f < n:
            assert np.array_equal(J1, G["J1"][f])
        assert np.array_equal(r, G["r"][f])
        assert c == pytest.approx(G["chi2"][f], rel=1e-15, abs=0)


def test_graph_chi2_matches_reference(oracle):
    G = golden("factor_eval.npz")
    n = len(G["pa"])
    states = np.vstack([G["pa"], G["pb"]])
    fa = np.concatenate([np.arange(n), np.arange(n)]).astype(np.int32)
    fb = np.concatenate([np.arange(n, 2 * n), -np.ones(n)]).astype(np.int32)
    c = oracle.chi2(states, fa, fb, np.vstack([G["z"], G["z"]]), np.vstack([G["W"], G["W"]]))
    assert c == pytest.approx(float(G["graph_chi2"][0]), rel=1e-13)


def test_normal_equations_match_reference(oracle):
    G = golden("normal_eq_10.npz")
    arr = datasets.random_pose_graph(10, 5, 7)
    A, B = oracle.normal_equations(arr[0], *arr[1:])
    assert np.allclose(np.triu(A), G["A_upper"], rtol=1e-12, atol=1e-9)
    assert np.allclose(B, G["B"], rtol=1e-12, atol=1e-9)


def test_m3500_batch_10_iterations(oracle):
    """Config 1: chi^2 per iteration and all 3500 final states (SURVEY.md §6 sequence)."""
    G = golden("m3500_batch.npz")
    chi2, st = oracle.iterate(datasets.m3500_batch(), 10)
    assert np.allclose(chi2, G["chi2"], rtol=1e-8)
    assert chi2[0] == pytest.approx(1283333.829603461, rel=1e-12)
    assert chi2[10] == pytest.approx(69.143589113, rel=1e-8)
    assert np.max(np.abs(st - G["final_states"])) < 1e-7


def test_m3500_first_step_states_and_dx(oracle):
    G = golden("m3500_batch.npz")
    arr = datasets.m3500_batch()
    st, dx, stats = oracle.batch_step(*arr)
    assert np.max(np.abs(st - G["states_after_1"])) < 1e-8
    assert np.max(np.abs(dx - G["dx_1"])) < 1e-8
    assert stats[0] > 0


@pytest.mark.parametrize("K", [6, 24, 60])
def test_lattice_batch(oracle, lib, K):
    G = golden(f"lattice_{K}.npz")
    iters = len(G["chi2"]) - 1
    chi2, st = oracle.iterate(lib.lattice_arrays(K), iters)
    assert np.allclose(chi2, G["chi2"], rtol=1e-8)
    assert np.max(np.abs(st - G["final_states"])) < 1e-7


@pytest.mark.parametrize("seed,shape", list(enumerate(((12, 6), (80, 60), (400, 350), (1500, 900)))))
def test_random_graphs_full_information(oracle, seed, shape):
    G = golden(f"random_{seed}.npz")
    chi2, st = oracle.iterate(datasets.random_pose_graph(shape[0], shape[1], seed), 3)
    assert np.allclose(chi2, G["chi2"], rtol=1e-7)
    assert np.max(np.abs(st - G["final_states"])) < 1e-6


def test_tutorial_batch_mode(oracle):
    """examples/aprilsam_tutorial.c in --batch_update_only mode: one batch step per added pose."""
    G = golden("tutorial_batch.npz")
    import math
    Wodo = np.diag([100.0, 100.0, 1.0 / math.radians(1) ** 2]).reshape(9)
    states = np.zeros((0, 3)); fa, fb, z, W = [0], [-1], [[0, 0, 0]], [datasets.PRIOR_W]
    for k in range(6):
        states = np.vstack([states, [k, 0, 0]])
        if k:
            fa.append(k - 1); fb.append(k); z.append([1, 0, 0]); W.append(Wodo)
        if k == 5:
            fa.append(0); fb.append(5); z.append([5, 1, 0]); W.append(Wodo)
        arr = (states, np.array(fa, np.int32), np.array(fb, np.int32), np.array(z, float), np.array(W))
        states, _, _ = oracle.batch_step(*arr)
        assert oracle.chi2(states, *arr[1:]) == pytest.approx(float(G["chi2"][k]), rel=1e-7, abs=1e-9)
        assert np.max(np.abs(states - G[f"states_{k}"])) < 1e-8
    assert G["chi2"][5] == pytest.approx(7.805041, abs=1e-6)


def test_oracle_matches_live_reference(oracle, reflib):
    """When oracle/_ref travelled: restatement vs the unmodified reference on a fresh seeded graph."""
    arr = datasets.random_pose_graph(200, 150, 11)
    g = reflib.new_graph(); g.build_from_arrays(*arr); p = reflib.new_param()
    ref = [g.chi2()]
    for _ in range(3):
        g.cholesky(p); ref.append(g.chi2())
    rst = g.states()
    p.destroy(); g.destroy()
    chi2, st = oracle.iterate(arr, 3)
    assert np.allclose(chi2, ref, rtol=1e-8)
    assert np.max(np.abs(st - rst)) < 1e-7


def test_information_matrices_not_symmetric_as_given(oracle):
    """W as the reference's text loader leaves it for correlated information (upper triangle filled, lower zero,
    examples/aprilsam_demo.c:73-75).  The reference accumulates only the upper triangle of its ORDERED matrix with W as given
    (aprilsam.c:171): its normal equations depend on its elimination order, which the fixture holds.  With that order the
    oracle reproduces the reference; with any other it must NOT (otherwise the fixture would not discriminate)."""
    G = golden("asym_batch.npz")
    arr = (G["states"], G["fa"], G["fb"], G["z"], G["W"])
    W = G["W"]
    assert np.any(W[:, 1] != W[:, 3]) and np.all(W[G["fb"] >= 0][:, [3, 6, 7]] == 0)
    chi2, st = oracle.iterate(arr, 3, order=G["ordering"])
    assert np.allclose(chi2, G["chi2"], rtol=1e-9)
    assert np.max(np.abs(st - G["states_after"][-1])) < 1e-8
    s1, dx1, _ = oracle.batch_step(G["states"], *arr[1:], order=G["ordering"])
    assert np.max(np.abs(dx1 - G["dx"][0])) < 1e-8
    # the oracle's own order: a different system
    s2, dx2, _ = oracle.batch_step(G["states"], *arr[1:])
    assert np.max(np.abs(dx2 - G["dx"][0])) > 1e-4
