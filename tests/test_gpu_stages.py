"""Stage-level device parity (plan items 1-2): what the HIP linearisation and the gather assembly
produce, compared with the reference's own per-factor J / r (tests/golden/factor_eval.npz, from
april_graph_xyt.c:62-124 and april_graph_xytpos.c:63-102) and its assembled normal equations
(tests/golden/normal_eq_10.npz: param->A un-permuted and param->B of aprilsam.c:159-204)."""
import ctypes as C
import os

import numpy as np
import pytest

from aprilsam_amd import datasets, host

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def lib():
    L = host.SolverLib()
    if L.device_count() < 1:
        pytest.fail("no HIP device visible")
    L.dll.aprilsam_amd_debug_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    return L


def _stage(lib, g, p, what, n):
    out = np.zeros(n)
    rc = lib.dll.aprilsam_amd_debug_stage(C.cast(g.ptr, C.c_void_p), C.cast(p.ptr, C.c_void_p), what, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0, rc
    return out


def test_linearisation_blocks_match_the_reference_jacobians(lib):
    """400 xyt factors between seeded random poses (incl. headings next to +-pi) and 400 xytpos priors: the device's
    J^T W J / J^T W r blocks against the products of the reference's J0, J1, r and W, factor by factor"""
    G = np.load(os.path.join(GOLD, "factor_eval.npz"))
    n = len(G["pa"])
    states = np.vstack([G["pa"], G["pb"]])
    fa = np.concatenate([np.arange(n), np.arange(n)]).astype(np.int32)
    fb = np.concatenate([np.arange(n, 2 * n), -np.ones(n)]).astype(np.int32)
    z = np.vstack([G["z"], G["z"]]); W = np.vstack([G["W"], G["W"]])
    g = lib.new_graph(); g.build_from_arrays(states, fa, fb, z, W); p = lib.new_param()
    H = _stage(lib, g, p, 0, 33 * 2 * n).reshape(2 * n, 33)
    worst = 0.0
    for f in range(2 * n):
        J0 = G["J0"][f].reshape(3, 3); Wf = W[f].reshape(3, 3); r = G["r"][f]
        want = [J0.T @ Wf @ J0]
        if f < n:
            J1 = G["J1"][f].reshape(3, 3)
            want += [J0.T @ Wf @ J1, J1.T @ Wf @ J1, J0.T @ Wf @ r, J1.T @ Wf @ r]
        else:
            want += [np.zeros((3, 3)), np.zeros((3, 3)), J0.T @ Wf @ r, np.zeros(3)]
        # (the library keeps the reference's upper triangle of the diagonal blocks, aprilsam.c:171, mirrored)
        want[0] = np.triu(want[0]) + np.triu(want[0], 1).T
        want[2] = np.triu(want[2]) + np.triu(want[2], 1).T
        got = [H[f, 0:9].reshape(3, 3), H[f, 9:18].reshape(3, 3), H[f, 18:27].reshape(3, 3), H[f, 27:30], H[f, 30:33]]
        for a, b in zip(got, want):
            scale = max(1.0, float(np.max(np.abs(b))))
            worst = max(worst, float(np.max(np.abs(a - b))) / scale)
    assert worst < 1e-12, worst
    # and the chi^2 kernel on the same graph (0.5 on xyt only, april_graph.c:86-93)
    assert g.chi2() == pytest.approx(float(G["graph_chi2"][0]), rel=1e-13)
    p.destroy(); g.destroy()


def test_assembled_normal_equations_match_the_reference(lib):
    """the assembly's destination sums, mapped back to node coordinates, against the reference's param->A / param->B"""
    G = np.load(os.path.join(GOLD, "normal_eq_10.npz"))
    arr = datasets.random_pose_graph(10, 5, 7)
    assert np.array_equal(arr[0], G["lp"])
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    out = _stage(lib, g, p, 1, 9 * 100 + 30)
    A = out[:900].reshape(30, 30); B = out[900:]
    assert np.allclose(A, A.T, rtol=0, atol=0)
    assert np.allclose(np.triu(A), G["A_upper"], rtol=1e-12, atol=1e-9)
    assert np.allclose(B, G["B"], rtol=1e-12, atol=1e-9)
    p.destroy(); g.destroy()


@pytest.mark.parametrize("leaf", [1, 4, 16])
def test_assembled_normal_equations_do_not_depend_on_the_plan(lib, leaf):
    """the same export under different nested-dissection leaf sizes (different fronts, slots and source lists) against the
    C oracle's normal equations on a graph with full information matrices"""
    from tests.support.oracle_binding import Oracle
    arr = datasets.random_pose_graph(60, 45, 11)
    Ao, Bo = Oracle().normal_equations(arr[0], *arr[1:])
    lib.set_option("leaf_nodes", leaf)
    try:
        g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
        out = _stage(lib, g, p, 1, 9 * 3600 + 180)
        A = out[:32400].reshape(180, 180); B = out[32400:]
        assert np.allclose(np.triu(A), np.triu(Ao), rtol=1e-12, atol=1e-9)
        assert np.allclose(B, Bo, rtol=1e-12, atol=1e-9)
        p.destroy(); g.destroy()
    finally:
        lib.set_option("leaf_nodes", 16)
