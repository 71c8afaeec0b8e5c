"""A size-independent parity property at BASELINE.json's full sizes: the dx the HIP path returns satisfies the reference's
normal equations (J'WJ + lambda I) dx = J'W r at the step's linearisation point, to rounding level -- checked matrix-free
in numpy (tests/support/normal_eq.py, pinned on the CPU against the oracle and the unmodified reference by
tests/test_normal_eq.py).  Unlike a chi^2 trace recorded by an earlier build, this needs nothing but the inputs: it is
the check of the 10^6-pose lattice (config 5), where neither the reference nor the oracle can run."""
import numpy as np
import pytest

from aprilsam_amd import datasets
from tests.support.normal_eq import normal_equation_residual

pytestmark = pytest.mark.gpu
LAM = 1e-4
RES_RTOL = 1e-10         # observed: 1e-14 .. 4e-12 (the reference itself: 2e-15, tests/test_normal_eq.py)


def _check_api_steps(lib, arr, iters):
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    worst = 0.0
    for _ in range(iters):
        g.cholesky(p)
        assert p.stats()["not_spd"] == 0
        out = normal_equation_residual(g.l_points(), arr[1], arr[2], arr[3], arr[4], g.deltas(), LAM)
        assert out["rel_max"] < RES_RTOL, out
        worst = max(worst, out["rel_max"])
    p.destroy(); g.destroy()
    return worst


def test_m3500_api_steps_satisfy_the_normal_equations(lib):
    _check_api_steps(lib, datasets.m3500_batch(), 3)


@pytest.mark.parametrize("K", [24, 100, 316])
def test_lattice_api_steps_satisfy_the_normal_equations(lib, K):
    """K = 316 is config 4 (99 856 poses): every kernel of the multi-workgroup path takes part"""
    _check_api_steps(lib, lib.lattice_arrays(K), 2)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_graphs_satisfy_the_normal_equations(lib, seed):
    _check_api_steps(lib, datasets.with_prior(*datasets.random_pose_graph(1500, 1200, seed)), 2)


@pytest.mark.parametrize("K,iters", [(316, 2), (1000, 1), (1000, 3)])
def test_resident_lattice_steps_satisfy_the_normal_equations(lib, K, iters):
    """the path bench.py times (resident_begin / steps / end), up to config 5's 10^6 poses on one GPU; resident_end leaves the LAST
    iteration's linearisation point and dx in the node objects"""
    arr = lib.lattice_arrays(K)
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    assert lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr) == 0
    lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, iters, 0)
    assert lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr) == 0
    assert lib.dll.aprilsam_amd_resident_end(g.ptr, p.ptr) == 0
    lp, dx, st = g.l_points(), g.deltas(), g.states()
    if iters == 1:
        assert np.array_equal(lp, arr[0])
    out = normal_equation_residual(lp, arr[1], arr[2], arr[3], arr[4], dx, LAM)
    assert out["rel_max"] < RES_RTOL, out
    d = st - lp - dx
    d[:, 2] = (d[:, 2] + np.pi) % (2 * np.pi) - np.pi
    assert np.max(np.abs(d)) < 1e-9
    # negative control at full size: one pose's x off by 1e-6
    bad = dx.copy(); bad[len(bad) // 3, 0] += 1e-6
    assert normal_equation_residual(lp, arr[1], arr[2], arr[3], arr[4], bad, LAM)["rel_max"] > 100 * max(out["rel_max"], 1e-13)
    p.destroy(); g.destroy()


def test_resident_loop_leaves_the_node_objects_as_the_same_number_of_api_calls_would(lib):
    """l_point after K iterations is the point the K-th step was linearised at (aprilsam.c:131-135) -- the point the factorisation kept for
    april_graph_cholesky_inc belongs to -- not the final state; state and delta_X likewise"""
    arr = datasets.m3500_batch()
    ga = lib.new_graph(); ga.build_from_arrays(*arr); pa = lib.new_param()
    for _ in range(3):
        ga.cholesky(pa)
    gr = lib.new_graph(); gr.build_from_arrays(*arr); pr = lib.new_param()
    assert lib.dll.aprilsam_amd_resident_begin(gr.ptr, pr.ptr) == 0
    lib.dll.aprilsam_amd_resident_steps(gr.ptr, pr.ptr, 2, 0)
    lib.dll.aprilsam_amd_resident_steps(gr.ptr, pr.ptr, 1, 0)           # (the last step of the last call counts)
    assert lib.dll.aprilsam_amd_resident_sync(gr.ptr, pr.ptr) == 0
    assert lib.dll.aprilsam_amd_resident_end(gr.ptr, pr.ptr) == 0
    assert np.max(np.abs(gr.l_points() - ga.l_points())) < 1e-12
    assert np.max(np.abs(gr.deltas() - ga.deltas())) < 1e-12
    assert np.max(np.abs(gr.states() - ga.states())) < 1e-12
    assert np.max(np.abs(gr.l_points() - gr.states())) > 1e-6           # (it is NOT the final state)
    # ... and an incremental step on top of either lands in the same place
    for g, p in ((ga, pa), (gr, pr)):
        n = g.n_nodes
        g.add_factor_xyt(10, n - 10, [0.3, -0.2, 0.05], np.diag([50.0, 50.0, 500.0]))
        g.cholesky_inc(p)
    assert np.max(np.abs(gr.states() - ga.states())) < 1e-9
    for g, p in ((ga, pa), (gr, pr)):
        p.destroy(); g.destroy()
