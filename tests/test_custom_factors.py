"""factor types neither library knows (tests/support/custom_factor.c, type tags 77 / 78) go through
their own eval() function pointer on the host; their J^T W J blocks join the device assembly.  Golden =
the unmodified reference on the same scenario (oracle/gen_golden.py --custom)."""
import numpy as np
import pytest

from tests.conftest import golden
from tests.support import custom_scenario


@pytest.fixture(scope="module")
def custom(tmp_path_factory):
    return custom_scenario.build_custom_lib(str(tmp_path_factory.mktemp("custom")))


def _check(out, G, tol):
    assert int(out["n_factors"]) == int(G["n_factors"])
    assert np.max(np.abs(out["chi2"] - G["chi2"]) / G["chi2"]) < tol, (out["chi2"], G["chi2"])
    for k in ("batch_states", "inc_states_0", "inc_states_1"):
        assert np.max(np.abs(out[k] - G[k])) < tol, k


def test_reference_reproduces_the_custom_factor_golden(reflib, custom):
    """pins the fixture (and the helper) against the live reference where it is available"""
    _check(custom_scenario.run(reflib, custom), golden("custom_factors.npz"), 1e-12)


def test_reference_reproduces_the_three_pose_factor_golden(reflib, custom):
    """... and the fixture with factors of THREE poses (factor->nnodes == 3, aprilsam.c:159-192 is generic over it)"""
    _check(custom_scenario.run(reflib, custom, triples=8), golden("custom_factors3.npz"), 1e-12)


def test_custom_factor_helper_is_consistent(custom):
    """the helper's objects have the public layout and a working destroy entry"""
    import ctypes as C
    from aprilsam_amd import abi
    z = (C.c_double * 2)(0.3, -0.2); W = (C.c_double * 4)(2.0, 0.1, 0.1, 3.0)
    f = custom.custom_xy_create(0, 1, z, W)
    assert f.contents.type == 77 and f.contents.nnodes == 2 and f.contents.length == 2
    assert C.sizeof(abi.Factor) == 104
    C.CFUNCTYPE(None, C.POINTER(abi.Factor))(f.contents.destroy)(f)


@pytest.mark.gpu
def test_custom_factors_match_reference_golden(lib, custom):
    out = custom_scenario.run(lib, custom)
    _check(out, golden("custom_factors.npz"), 1e-6)


@pytest.mark.gpu
def test_factors_with_three_poses_match_reference_golden(lib, custom):
    """round 4: a foreign factor with more than two nodes is packed as the clique of its node pairs (each pair carries one
    off-diagonal block J_i^T W J_j; a node's diagonal block and right-hand-side segment ride on its first pair) -- 8 such factors
    in the batch graph, one more in each of the two incremental steps, against the unmodified reference; param->factor_num
    keeps counting GRAPH factors"""
    G = golden("custom_factors3.npz")
    out = custom_scenario.run(lib, custom, triples=8)
    _check(out, G, 1e-6)


@pytest.mark.gpu
def test_resident_api_refuses_host_evaluated_factors(lib, custom):
    import ctypes as C
    g = lib.new_graph(); p = lib.new_param()
    for i in range(3):
        g.add_node_xyt([float(i), 0.0, 0.0])
    g.add_factor_xytpos(0, [0, 0, 0], np.diag([1e4, 1e4, 1e3]))
    g.add_factor_xyt(0, 1, [1, 0, 0], np.eye(3)); g.add_factor_xyt(1, 2, [1, 0, 0], np.eye(3))
    lib._add_factor(g.ptr, custom.custom_heading_create(2, 0.1, 100.0))
    assert lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr) == -4
    g.cholesky(p)                                   # the API path works
    assert np.isfinite(g.states()).all() and abs(g.states()[2, 2] - 0.1) < 0.1
    p.destroy(); g.destroy()
