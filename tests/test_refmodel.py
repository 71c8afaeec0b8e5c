"""Host logic of the incremental path (no GPU): the restated reference elimination order, block elimination
tree and solve_node bookkeeping, checked against the unmodified reference (oracle/_ref) stepping the demo
scenario pose by pose.  The x fed to the model comes from the oracle's exact solve of the incremental system —
which also re-establishes, step by step, the fact the design rests on: on the poses it touches, the
reference's incremental result IS the exact solution of that system."""
import ctypes as C

import numpy as np
import pytest

from aprilsam_amd import datasets, harness

ip = C.POINTER(C.c_int)
dp = C.POINTER(C.c_double)


def _i(a):
    return a.ctypes.data_as(ip)


@pytest.mark.parametrize("case", ["m3500", "random0", "random1", "random2", "lattice24", "lattice40", "prefix232", "prefix1088"])
def test_reference_order_and_tree_restated_exactly(lib, reflib, case):
    if case == "m3500":
        arr = datasets.m3500_batch()
    elif case.startswith("lattice"):
        arr = lib.lattice_arrays(int(case[7:]))
    elif case.startswith("prefix"):           # the graph of the M3500 demo at one of its batch fall-backs
        n = int(case[6:])
        st, fa, fb, z, W = datasets.m3500_arrays(); m = np.maximum(fa, fb) < n
        arr = datasets.with_prior(st[:n], fa[m], fb[m], z[m], W[m])
    else:
        seed = int(case[-1]); arr = datasets.random_pose_graph(*((12, 6), (80, 60), (400, 350))[seed], seed)
    st, fa, fb = arr[0], np.ascontiguousarray(arr[1], np.int32), np.ascontiguousarray(arr[2], np.int32)
    N = len(st)
    g = reflib.new_graph(); g.build_from_arrays(*arr); p = reflib.new_param(); g.cholesky(p)
    ro = np.zeros(N, np.int32); rp = np.zeros(N, np.int32)
    reflib.dll.rs_param_ordering(p.ptr, _i(ro)); reflib.dll.rs_tree_parents(p.ptr, _i(rp))
    mo = np.zeros(N, np.int32); mp = np.zeros(N, np.int32)
    lib.dll.aprilsam_amd_reference_order(N, len(fa), _i(fa), _i(fb), _i(mo), _i(mp))
    assert np.array_equal(ro, mo) and np.array_equal(rp, mp)
    p.destroy(); g.destroy()


def test_restated_order_equals_the_order_in_the_asymmetric_w_fixture(lib):
    """the library orients factors with an asymmetric W by the reference's elimination order: the restated order must equal the one the
    reference used when the fixture was made (graph with priors on every fourth pose)"""
    from tests.conftest import golden
    G = golden("asym_batch.npz")
    fa, fb = np.ascontiguousarray(G["fa"], np.int32), np.ascontiguousarray(G["fb"], np.int32)
    N = len(G["states"])
    mo = np.zeros(N, np.int32); mp = np.zeros(N, np.int32)
    lib.dll.aprilsam_amd_reference_order(N, len(fa), _i(fa), _i(fb), _i(mo), _i(mp))
    assert np.array_equal(mo, G["ordering"])


def test_incremental_bookkeeping_follows_the_reference_step_by_step(lib, reflib, oracle):
    """first 260 poses of the M3500 demo (contains the first batch fall-back at 232 nodes)"""
    NST = 260
    states, fa, fb, z, W = datasets.m3500_arrays()
    by_pose = [[] for _ in range(NST)]
    for k in range(len(fa)):
        m = max(int(fa[k]), int(fb[k]))
        if m < NST:
            by_pose[m].append(k)
    lib.dll.aprilsam_amd_refmodel_create.restype = C.c_void_p
    lib.dll.aprilsam_amd_refmodel_solve_visit.argtypes = [C.c_void_p, dp, C.c_double, C.c_double, ip]
    lib.dll.aprilsam_amd_refmodel_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, ip]
    lib.dll.aprilsam_amd_refmodel_inc_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, ip]
    lib.dll.aprilsam_amd_refmodel_get.argtypes = [C.c_void_p, ip, ip, ip]
    lib.dll.aprilsam_amd_refmodel_destroy.argtypes = [C.c_void_p]
    M = C.c_void_p(lib.dll.aprilsam_amd_refmodel_create())
    g = reflib.new_graph(); p = reflib.new_param()
    cfa, cfb, cz, cW = [], [], [], []
    batch_nodes = 0
    fallbacks = []
    for k in range(NST):
        g.add_node_xyt(states[k])
        if k == 0:
            g.add_factor_xytpos(0, [0, 0, 0], datasets.PRIOR_W); cfa.append(0); cfb.append(-1); cz.append([0, 0, 0]); cW.append(datasets.PRIOR_W)
        for f in (by_pose[k] if k else []):
            a, b = int(fa[f]), int(fb[f])
            if abs(a - b) == 1:
                if a < b:
                    g.set_state(b, harness._xyt_mul(g.states_of(a), z[f]), relinearize=True)
                else:
                    g.set_state(a, harness._xyt_mul(g.states_of(b), harness._xyt_inv(z[f])), relinearize=True)
            g.add_factor_xyt(a, b, z[f], W[f]); cfa.append(a); cfb.append(b); cz.append(z[f]); cW.append(W[f])
        afa = np.array(cfa, np.int32); afb = np.array(cfb, np.int32)
        N = k + 1
        if k == 0:
            g.cholesky(p); batch_nodes = 1
            lib.dll.aprilsam_amd_refmodel_batch(M, N, len(afa), _i(afa), _i(afb))
            continue
        st_before, lp, dx_before = g.states(), g.l_points(), g.deltas()
        p.c.batch_time = 1e300; bt = p.c.batch_time
        g.cholesky_inc(p)
        fell_back = p.c.batch_time != bt
        # ---- the model, fed with the exact solution of the incremental system
        naff = lib.dll.aprilsam_amd_refmodel_inc_begin(M, N, len(afa), _i(afa), _i(afb))
        lam = np.where(np.arange(N) < batch_nodes, 1e-4, 0.0)
        x = np.ascontiguousarray(oracle.solve_system(lp, lp, afa, afb, np.array(cz), np.array(cW), lam))
        visited = np.zeros(N, np.int32)
        so = lib.dll.aprilsam_amd_refmodel_solve_visit(M, x.ctypes.data_as(dp), 0.1, 0.1, _i(visited))
        if fell_back:
            fallbacks.append(N)
            assert so > 100                                  # the model triggers the same fall-back (nthreshold 100)
            batch_nodes = N
            lib.dll.aprilsam_amd_refmodel_batch(M, N, len(afa), _i(afa), _i(afb))
            continue
        assert naff == reflib.dll.rs_tree_naffected(p.ptr)
        assert so == reflib.dll.rs_tree_start_over(p.ptr)
        rp = np.zeros(N, np.int32); rc = np.zeros(N, np.int32); rr = np.zeros(N, np.int32)
        reflib.dll.rs_tree_parents(p.ptr, _i(rp)); reflib.dll.rs_tree_labels(p.ptr, _i(rc), _i(rr))
        mp = np.zeros(N, np.int32); mc = np.zeros(N, np.int32); mr = np.zeros(N, np.int32)
        lib.dll.aprilsam_amd_refmodel_get(M, _i(mp), _i(mc), _i(mr))
        assert np.array_equal(rp, mp) and np.array_equal(rc, mc) and np.array_equal(rr, mr)
        # states: updated poses = l_point + x; everything else untouched; delta_X of visited poses = x
        st_after, dx_after = g.states(), g.deltas()
        upd = visited == 2
        pred = lp + x; pred[:, 2] = [oracle.mod2pi(v) for v in pred[:, 2]]
        assert np.max(np.abs(pred[upd] - st_after[upd]), initial=0) < 1e-9
        assert np.array_equal(st_after[~upd], st_before[~upd])
        vis = visited > 0
        assert np.max(np.abs(dx_after[vis] - x[vis]), initial=0) < 1e-9
        assert np.array_equal(dx_after[~vis], dx_before[~vis])
    assert fallbacks == [232]                                # first fall-back at 232 nodes
    lib.dll.aprilsam_amd_refmodel_destroy(M)
    p.destroy(); g.destroy()


def test_incrementally_maintained_tree_equals_a_full_recomputation(lib):
    """RefModel keeps the block elimination tree up to date by merging root paths per new edge (refmodel.cpp insert_edge)
    instead of recomputing it from the whole graph at every incremental step: random growth -- new poses with odometry, loop
    closures to old poses, several factors per step, duplicates, factors between two OLD poses, priors -- must leave exactly
    the tree (and children lists) a recomputation gives."""
    d = lib.dll
    d.aprilsam_amd_refmodel_create.restype = C.c_void_p
    d.aprilsam_amd_refmodel_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, ip]
    d.aprilsam_amd_refmodel_inc_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, ip]
    d.aprilsam_amd_refmodel_check.argtypes = [C.c_void_p]
    d.aprilsam_amd_refmodel_solve_visit.argtypes = [C.c_void_p, dp, C.c_double, C.c_double, ip]
    d.aprilsam_amd_refmodel_destroy.argtypes = [C.c_void_p]
    for seed in range(6):
        rng = np.random.default_rng(seed)
        n0 = int(rng.integers(1, 60))
        st, fa, fb, z, W = datasets.random_pose_graph(max(n0, 2), int(rng.integers(0, 40)), seed)
        fa, fb = list(map(int, fa)), list(map(int, fb))
        N = len(st)
        M = C.c_void_p(d.aprilsam_amd_refmodel_create())
        a = np.array(fa, np.int32); b = np.array(fb, np.int32)
        d.aprilsam_amd_refmodel_batch(M, N, len(a), _i(a), _i(b))
        assert d.aprilsam_amd_refmodel_check(M) == 0
        for step in range(120):
            for _ in range(int(rng.integers(0, 3))):
                fa.append(N - 1); fb.append(N); N += 1                       # a new pose with its odometry factor
            for _ in range(int(rng.integers(0, 4))):
                u, v = int(rng.integers(0, N)), int(rng.integers(0, N))
                if u != v:
                    fa.append(u); fb.append(v)                               # anything to anything: closures, old-old, duplicates
            if rng.random() < 0.1:
                fa.append(int(rng.integers(0, N))); fb.append(-1)            # a prior
            a = np.array(fa, np.int32); b = np.array(fb, np.int32)
            d.aprilsam_amd_refmodel_inc_begin(M, N, len(a), _i(a), _i(b))
            assert d.aprilsam_amd_refmodel_check(M) == 0, (seed, step)
            x = np.zeros(3 * N); vis = np.zeros(N, np.int32)
            d.aprilsam_amd_refmodel_solve_visit(M, x.ctypes.data_as(dp), 0.1, 0.1, _i(vis))     # clears the labels like a real step
        d.aprilsam_amd_refmodel_destroy(M)
