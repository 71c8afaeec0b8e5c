"""Information matrices that are NOT symmetric as given (what the reference's text loader produces for correlated information,
examples/aprilsam_demo.c:73-75: W[1], W[2], W[5] filled, W[3], W[6], W[7] zero).  The reference uses W as given and accumulates only
the upper triangle of its ORDERED normal equations (aprilsam.c:162,171,520): the diagonal blocks are the mirrored
upper triangles of J'WJ, and the off-diagonal block of a factor (a, b) is J_a'W J_b when a is eliminated first, (J_b'W J_a)' otherwise.
The library reproduces that with the reference's own order (csrc/refmodel.cpp) behind a bit of the per-factor swap byte
(solver_context.inc.h: orient_asymmetric; kernels.hip.h: linearise_factor).  Fixtures: oracle/gen_golden.py --asym (unmodified
reference); bars: chi^2 <= 1e-6 relative, states <= 1e-6."""
import numpy as np
import pytest

from aprilsam_amd import harness
from tests.conftest import golden
from tests.support import asym_scenarios

pytestmark = pytest.mark.gpu
CHI2_RTOL = 1e-6
STATE_ATOL = 1e-6


def _arrays(G):
    return G["states"], G["fa"], G["fb"], G["z"], G["W"]


def test_fixture_inputs_are_what_the_scenario_module_builds():
    G = golden("asym_batch.npz")
    for a, b in zip(asym_scenarios.batch_graph(), _arrays(G)):
        assert np.array_equal(a, b)
    G = golden("asym_inc_demo.npz")
    for a, b in zip(asym_scenarios.growth_graph(), _arrays(G)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("opts", [dict(), dict(linearize_staged_min=0), dict(small_lds_kb=0), dict(use_graph=0), dict(trust_factor_cache=1)])
def test_batch_steps_match_the_reference_golden(lib, opts):
    G = golden("asym_batch.npz")
    saved = {k: lib.get_option(k) for k in opts}
    try:
        for k, v in opts.items():
            lib.set_option(k, v)
        g = lib.new_graph(); g.build_from_arrays(*_arrays(G)); p = lib.new_param()
        chi2 = [g.chi2()]
        for it in range(3):
            g.cholesky(p); chi2.append(g.chi2())
            assert np.max(np.abs(g.states() - G["states_after"][it])) < STATE_ATOL
            assert np.max(np.abs(g.deltas() - G["dx"][it])) < STATE_ATOL
        st = p.stats()
        assert st["not_spd"] == 0 and st["error_code"] == 0
        assert np.max(np.abs(np.array(chi2) - G["chi2"]) / G["chi2"]) < CHI2_RTOL
        p.destroy(); g.destroy()
    finally:
        for k, v in saved.items():
            lib.set_option(k, v)


def test_the_symmetrised_matrices_give_a_different_answer(lib):
    """the fixture discriminates: with (W + W') / 2 in place of W as given the first step differs from the reference's by far more
    than the bar -- a library that symmetrised W, or ignored the reference's order, would fail the test above"""
    G = golden("asym_batch.npz")
    st, fa, fb, z, W = _arrays(G)
    Ws = W.reshape(-1, 3, 3); Ws = ((Ws + Ws.transpose(0, 2, 1)) / 2).reshape(-1, 9)
    g = lib.new_graph(); g.build_from_arrays(st, fa, fb, z, Ws); p = lib.new_param()
    g.cholesky(p)
    assert np.max(np.abs(g.deltas() - G["dx"][0])) > 1e-4
    p.destroy(); g.destroy()


def test_resident_iterations_on_asymmetric_matrices(lib):
    G = golden("asym_batch.npz")
    g = lib.new_graph(); g.build_from_arrays(*_arrays(G)); p = lib.new_param()
    chi2, _ = g.batch_resident(p, 3)
    assert np.max(np.abs(chi2 - G["chi2"]) / G["chi2"]) < CHI2_RTOL
    assert np.max(np.abs(g.states() - G["states_after"][-1])) < STATE_ATOL
    p.destroy(); g.destroy()


def test_edit_in_place_from_symmetric_to_asymmetric_and_back(lib):
    """a warm call notices that a packed factor's W became asymmetric (the orientation bits follow), and that it became symmetric again"""
    G = golden("asym_batch.npz")
    st, fa, fb, z, W = _arrays(G)
    Ws = W.reshape(-1, 3, 3); Ws = ((Ws + Ws.transpose(0, 2, 1)) / 2).reshape(-1, 9)
    g = lib.new_graph(); g.build_from_arrays(st, fa, fb, z, Ws); p = lib.new_param()
    g.cholesky(p); g.cholesky(p)                        # warm: the next call would speculate on unchanged factors
    g.set_all_states(st, relinearize=True)
    g.set_all_W(W)
    g.cholesky(p)
    assert np.max(np.abs(g.deltas() - G["dx"][0])) < STATE_ATOL
    sym_first = lib.new_graph(); sym_first.build_from_arrays(st, fa, fb, z, Ws); q = lib.new_param(); sym_first.cholesky(q)
    g.set_all_states(st, relinearize=True)
    g.set_all_W(Ws)
    g.cholesky(p)
    assert np.max(np.abs(g.deltas() - sym_first.deltas())) < 1e-9
    q.destroy(); sym_first.destroy(); p.destroy(); g.destroy()


@pytest.mark.parametrize("opts", [dict(), dict(inc_update=0), dict(inc_one=0, inc_tail=0)])
def test_incremental_growth_with_loop_closures_matches_the_reference_golden(lib, opts):
    """800 poses of M3500 added one by one (examples/aprilsam_demo.c semantics) with loader-style correlations, priors on every fifth
    pose, a caller-made batch step every 250 poses: every new factor is oriented by the positions it enters at (aprilsam.c:393-396,520),
    every batch step re-orients all of them; loop closures with an asymmetric W cannot take the low-rank update (W = C C' does not
    exist) and are re-assembled (solver_inc.inc.h)."""
    G = golden("asym_inc_demo.npz")
    saved = {k: lib.get_option(k) for k in opts}
    try:
        for k, v in opts.items():
            lib.set_option(k, v)
        res = harness.run_demo(lib, _arrays(G), deterministic=True, record_states_every=100, batch_every=asym_scenarios.GROWTH_BATCH_EVERY)
    finally:
        for k, v in saved.items():
            lib.set_option(k, v)
    assert np.array_equal(res["was_batch"], G["was_batch"])
    rel = np.abs(res["chi2"] - G["chi2"]) / np.maximum(G["chi2"], 1e-9)
    assert rel.max() < CHI2_RTOL, (int(rel.argmax()), rel.max())
    for k, v in res["snaps"].items():
        assert np.max(np.abs(v - G[f"snap_{k}"])) < STATE_ATOL, k
    assert np.max(np.abs(res["final_states"] - G["final_states"])) < STATE_ATOL
