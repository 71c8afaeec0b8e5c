"""Parity tests proper: the HIP path, called through the C-ABI, against (i) golden vectors from the
unmodified reference, (ii) the CPU oracle on seeded inputs, (iii) the reference library itself when
oracle/_ref travelled, (iv) size-independent properties at BASELINE.json's full sizes.

Tolerances (BASELINE.json north_star): chi^2 within 1e-6 relative, node states within 1e-6 of the
reference CPU path.  What we actually observe is ~1e-11 (FP64 everywhere, different but valid
elimination order), so the tests assert tighter bounds where that is robust.
"""
import numpy as np
import pytest

from aprilsam_amd import datasets, harness
from tests.conftest import golden

pytestmark = pytest.mark.gpu
CHI2_RTOL = 1e-6      # the bar
STATE_ATOL = 1e-6


def run_batch(lib, arr, iters):
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    chi2 = [g.chi2()]
    snaps = []
    for _ in range(iters):
        g.cholesky(p); chi2.append(g.chi2()); snaps.append((g.states(), g.deltas(), g.l_points()))
    stats = p.stats()
    p.destroy(); g.destroy()
    return np.array(chi2), snaps, stats


def test_gpu_is_used(lib):
    assert lib.device_count() >= 1


def test_chi2_kernel_matches_reference(lib):
    G = golden("factor_eval.npz")
    n = len(G["pa"])
    g = lib.new_graph()
    g.build_from_arrays(np.vstack([G["pa"], G["pb"]]), np.concatenate([np.arange(n), np.arange(n)]),
                        np.concatenate([np.arange(n, 2 * n), -np.ones(n)]), np.vstack([G["z"], G["z"]]), np.vstack([G["W"], G["W"]]))
    assert g.chi2() == pytest.approx(float(G["graph_chi2"][0]), rel=1e-12)
    g.destroy()


def test_m3500_batch_matches_reference_golden(lib):
    """Config 2: 10 batch iterations on M3500: chi^2 per iteration, states after 1 and after 10."""
    G = golden("m3500_batch.npz")
    chi2, snaps, stats = run_batch(lib, datasets.m3500_batch(), 10)
    assert np.max(np.abs(chi2 - G["chi2"]) / G["chi2"]) < CHI2_RTOL
    assert np.max(np.abs(chi2 - G["chi2"]) / G["chi2"]) < 1e-8
    assert np.max(np.abs(snaps[0][0] - G["states_after_1"])) < STATE_ATOL
    assert np.max(np.abs(snaps[0][1] - G["dx_1"])) < STATE_ATOL
    assert np.max(np.abs(snaps[-1][0] - G["final_states"])) < STATE_ATOL
    assert stats["not_spd"] == 0 and stats["n_nodes"] == 3500 and stats["symbolic_reused"] == 1


@pytest.mark.parametrize("K", [6, 24, 60, 120])
def test_lattice_matches_reference_golden(lib, K):
    G = golden(f"lattice_{K}.npz")
    chi2, snaps, _ = run_batch(lib, lib.lattice_arrays(K), len(G["chi2"]) - 1)
    assert np.max(np.abs(chi2 - G["chi2"]) / G["chi2"]) < 1e-8
    assert np.max(np.abs(snaps[-1][0] - G["final_states"])) < STATE_ATOL


@pytest.mark.parametrize("seed,shape", list(enumerate(((12, 6), (80, 60), (400, 350), (1500, 900)))))
def test_random_full_information_graphs_match_reference_golden(lib, seed, shape):
    G = golden(f"random_{seed}.npz")
    chi2, snaps, _ = run_batch(lib, datasets.random_pose_graph(shape[0], shape[1], seed), 3)
    assert np.max(np.abs(chi2 - G["chi2"]) / G["chi2"]) < CHI2_RTOL
    assert np.max(np.abs(snaps[-1][0] - G["final_states"])) < STATE_ATOL


def test_tutorial_batch_mode_matches_reference_golden(lib):
    G = golden("tutorial_batch.npz")
    res = harness.run_tutorial(lib, batch_update_only=True)
    for k, (c, st) in enumerate(res):
        assert c == pytest.approx(float(G["chi2"][k]), rel=CHI2_RTOL, abs=1e-9)
        assert np.max(np.abs(st - G[f"states_{k}"])) < STATE_ATOL
    assert res[-1][0] == pytest.approx(7.805041, abs=1e-6)


@pytest.mark.parametrize("opts", [dict(small_lds_kb=0), dict(small_lds_kb=48), dict(small_lds_kb=156), dict(panel_mode=0, small_lds_kb=64),
                                  dict(small_threads=256), dict(small_threads=512), dict(tp_fronts=1, tp_lds_kb=8), dict(small_lds_kb=0, use_graph=0),
                                  dict(schur_first=1), dict(schur_first=1, small_lds_kb=48), dict(schur_first=1, persist=0, use_graph=0), dict(schur_first=8, leaf_nodes=40), dict(schur_first=0),
                                  dict(syrk_small_tiles=0, small_lds_kb=0), dict(syrk_small_tiles=1 << 30, small_lds_kb=0), dict(syrk_small_tiles=1 << 30, small_lds_kb=0, syrk_xcd_order=1),
                                  dict(syrk_xcd_order=1, small_lds_kb=0), dict(syrk_xcd_order=0, small_lds_kb=0),
                                  dict(syrk_pair_tiles=1, syrk_group=2, small_lds_kb=0), dict(syrk_pair_tiles=1, syrk_group=3, small_lds_kb=0), dict(syrk_pair_tiles=1, syrk_group=4, small_lds_kb=0, syrk_small_tiles=0),
                                  dict(syrk_pair_tiles=1, syrk_group=3, small_lds_kb=48, use_graph=0), dict(syrk_pair_tiles=0, small_lds_kb=0),
                                  dict(blk_backsolve=0, small_lds_kb=0), dict(blk_backsolve=0, small_lds_kb=48), dict(small_lds_kb=0, leaf_nodes=64), dict(small_lds_kb=0, leaf_nodes=4, use_graph=0),
                                  dict(leaf_nodes=4), dict(leaf_nodes=40), dict(use_graph=0), dict(device_timing=1), dict(pin_last=12), dict(trust_factor_cache=1),
                                  dict(linearize_staged_min=0), dict(persist=0), dict(persist_max_fronts=100000), dict(wave_backsolve=0), dict(wave_backsolve=0, persist=0)])
def test_every_kernel_path_agrees_with_oracle(lib, oracle, opts):
    """force the multi-workgroup big-front path (outer-block panels: diagonal block in LDS, row solves on the matrix cores, one wide
    update per 128 columns in both tile sizes and tile orders), small LDS budgets (more panel-mode and big fronts), panel mode off,
    other workgroup sizes and leaf sizes, the newest poses pinned into the root front, no hipGraph, k_linearize with the LDS-staged
    write-out, no / all-level multi-level launches, both back-substitution forms: same answers.  (Round 5 removed the kernel variants
    that earlier rounds had measured slower and retired -- per-panel forms, LDS-staged 128 x 128 update, tile assembly, side-stream
    look-ahead, per-pivot elimination -- together with their options.)"""
    saved = {k: lib.get_option(k) for k in opts}
    arr = datasets.random_pose_graph(700, 600, 21)
    oc, ost = oracle.iterate(arr, 2)
    try:
        for k, v in opts.items():
            lib.set_option(k, v)
        chi2, snaps, stats = run_batch(lib, arr, 2)
    finally:
        for k, v in saved.items():
            lib.set_option(k, v)
    assert np.max(np.abs(chi2 - oc) / oc) < 1e-8
    assert np.max(np.abs(snaps[-1][0] - ost)) < STATE_ATOL
    if "device_timing" in opts:
        assert stats["ms_dev_factor"] > 0


def test_root_front_ending_in_a_partial_outer_block(lib, oracle):
    """3 000 poses with 1 800 random loop closures: the root front owns 645 poses = 15 outer blocks of 128 columns + 15 columns, and it
    is the last array of the front pool.  The row solves and the chain of the wide back substitution used to stage the L blocks of a
    whole 128-column block -- for that last block 80 columns past the end of the front, i.e. of the pool: a memory fault when the
    over-read crossed the end of the allocation (round 4: found by a randomised sweep, `tools/stress_random_graphs.py`)"""
    arr = datasets.random_pose_graph(3000, 1800, 102)
    oc, ost = oracle.iterate(arr, 2)
    for opts in (dict(), dict(small_lds_kb=0)):
        with lib.options(**opts):
            chi2, snaps, stats = run_batch(lib, arr, 2)
        assert stats["max_front_rows"] > 1900
        assert np.max(np.abs(chi2 - oc) / oc) < 1e-8
        assert np.max(np.abs(snaps[-1][0] - ost)) < STATE_ATOL


def _star(n_leaves, seed):
    """hub 0 connected to every other pose: one separator (the hub), n_leaves children of the root front"""
    rng = np.random.default_rng(seed)
    ang = rng.uniform(-np.pi, np.pi, n_leaves); rad = rng.uniform(1, 20, n_leaves)
    st = np.vstack([[0, 0, 0], np.column_stack([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(-np.pi, np.pi, n_leaves)])])
    fa = np.zeros(n_leaves, np.int32); fb = np.arange(1, n_leaves + 1, dtype=np.int32)
    z = np.column_stack([st[1:, 0] + rng.normal(0, .1, n_leaves), st[1:, 1] + rng.normal(0, .1, n_leaves), st[1:, 2] + rng.normal(0, .02, n_leaves)])
    W = np.tile(np.diag([50.0, 50.0, 200.0]).reshape(9), (n_leaves, 1))
    return datasets.with_prior(st, fa, fb, z, W, first=True)


def _chain(n, seed):
    rng = np.random.default_rng(seed)
    st = np.column_stack([np.arange(n, dtype=float), rng.normal(0, .2, n), rng.normal(0, .05, n)])
    fa = np.arange(n - 1, dtype=np.int32); fb = fa + 1
    z = np.column_stack([np.ones(n - 1), np.zeros(n - 1), np.zeros(n - 1)])
    W = np.tile(np.diag([100.0, 100.0, 400.0]).reshape(9), (n - 1, 1))
    return datasets.with_prior(st, fa, fb, z, W, first=False)


@pytest.mark.parametrize("name,arr", [("star_3000", _star(3000, 1)), ("star_70", _star(70, 2)), ("chain_4000", _chain(4000, 3))])
def test_degenerate_tree_shapes_agree_with_oracle(lib, oracle, name, arr):
    """a front with thousands of children (more than the staged child records and many work-list refills) and a
    path graph (deep, thin elimination tree)"""
    oc, ost = oracle.iterate(arr, 2)
    chi2, snaps, _ = run_batch(lib, arr, 2)
    assert np.max(np.abs(chi2 - oc)) < 1e-8 * oc[0]           # (every leaf of the star can satisfy its single factor: chi^2 -> 0)
    assert np.max(np.abs(snaps[-1][0] - ost)) < STATE_ATOL


def test_side_effects_on_the_graph_follow_the_reference(lib):
    """l_point = state before the step, delta_X = dx, UID = index, param bookkeeping (aprilsam.c:283-288, 628)"""
    arr = datasets.random_pose_graph(50, 30, 9)
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    before = g.states()
    g.cholesky(p)
    assert np.array_equal(g.l_points(), before)
    after, dx = g.states(), g.deltas()
    exp = before + dx
    assert np.allclose(after[:, :2], exp[:, :2], rtol=0, atol=1e-15)
    assert np.allclose(np.sin(after[:, 2]), np.sin(exp[:, 2]), atol=1e-12) and np.all(after[:, 2] >= -np.pi) and np.all(after[:, 2] < np.pi)
    assert [g.node(i).UID for i in range(g.n_nodes)] == list(range(g.n_nodes))
    assert p.c.nreordering == 50 and p.c.factor_num == g.n_factors
    assert sorted(p.c.ordering[i] for i in range(50)) == list(range(50))
    assert not p.c.chol and not p.c.A and not p.c.tr       # reference-owned CPU state stays NULL
    p.destroy(); g.destroy()


@pytest.mark.parametrize("case", ["random_900", "m3500_panel_mode", "lattice120_big_path"])
def test_bitwise_reproducible(lib, case):
    """fixed summation order everywhere, including the paths that accumulate with L2 atomics (update columns of
    panel-mode fronts on M3500, chunked assembly of the multi-workgroup path): two runs give identical bits"""
    opts = {}
    if case == "random_900":
        arr = datasets.random_pose_graph(900, 800, 33)
    elif case == "m3500_panel_mode":
        arr = datasets.m3500_batch()
    else:
        arr = lib.lattice_arrays(120); opts = dict(small_lds_kb=0)
    try:
        for k, v in opts.items():
            lib.set_option(k, v)
        a = run_batch(lib, arr, 3)
        b = run_batch(lib, arr, 3)
    finally:
        if opts:
            lib.set_option("small_lds_kb", 156)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1][-1][0], b[1][-1][0])


def test_resident_iterations_equal_api_iterations(lib):
    arr = datasets.m3500_batch()
    chi2_api, snaps, _ = run_batch(lib, arr, 4)
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    chi2_res, ms = g.batch_resident(p, 4)
    st = g.states()
    assert np.array_equal(chi2_res, chi2_api)
    assert np.array_equal(st, snaps[-1][0])
    assert np.all(ms > 0)
    p.destroy(); g.destroy()


def test_not_positive_definite_leaves_states_untouched(lib):
    st = np.array([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0]])
    W = np.vstack([datasets.PRIOR_W, np.diag([-50.0, 10, 10]).reshape(9), np.diag([10.0, 10, 10]).reshape(9)])
    arr = (st, np.array([0, 0, 1], np.int32), np.array([-1, 1, 2], np.int32), np.array([[0, 0, 0], [1, 0.1, 0], [1, 0, 0.1]]), W)
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    g.cholesky(p)
    assert p.stats()["not_spd"] == 1
    assert np.array_equal(g.states(), st)
    p.destroy(); g.destroy()


def test_empty_graph_and_isolated_nodes(lib):
    g = lib.new_graph(); p = lib.new_param()
    g.cholesky(p)                                    # silent no-op (aprilsam.c:90-91)
    g.add_node_xyt([1, 2, 3])
    g.cholesky(p)                                    # nodes but no factors: still a no-op
    assert g.states().tolist() == [[1, 2, 3]]
    g.add_node_xyt([0.5, 0.5, 0.1])
    g.add_factor_xytpos(1, [0, 0, 0], datasets.PRIOR_W)
    g.cholesky(p)                                    # node 0 is isolated: only Tikhonov on its diagonal -> dx = 0
    st = g.states()
    assert st[0].tolist() == [1, 2, 3 - 2 * np.pi] or np.allclose(st[0], [1, 2, 3])
    assert np.max(np.abs(st[1])) < 1e-6
    p.destroy(); g.destroy()


def test_reference_built_graph_solved_by_our_library(lib, reflib):
    """The drop-in direction: graph objects created by the UNMODIFIED reference library are handed to
    libaprilsam_amd's april_graph_cholesky / april_graph_chi2 and the result is compared with the
    reference solving its own copy."""
    arr = datasets.m3500_batch()
    g_ref = reflib.new_graph(); g_ref.build_from_arrays(*arr); p_ref = reflib.new_param()
    g_mix = reflib.new_graph(); g_mix.build_from_arrays(*arr)            # reference objects ...
    p_ours = lib.new_param()
    for it in range(3):
        g_ref.cholesky(p_ref)
        lib.dll.april_graph_cholesky(g_mix.ptr, p_ours.ptr)              # ... solved by the HIP path
        c_ref = g_ref.chi2()
        c_ours = float(lib.dll.april_graph_chi2(g_mix.ptr))
        assert c_ours == pytest.approx(c_ref, rel=CHI2_RTOL)
        assert float(reflib.dll.april_graph_chi2(g_mix.ptr)) == pytest.approx(c_ref, rel=CHI2_RTOL)
        assert np.max(np.abs(g_mix.states() - g_ref.states())) < STATE_ATOL
    p_ours.destroy(); p_ref.destroy(); g_ref.destroy(); g_mix.destroy()


def test_lattice_100k_full_size_properties(lib):
    """Config 4 at full size (99 856 poses / 397 531 factors).  Oracle-independent properties:
    the generator reproduces the reference's chi^2_0, one step lands on the reference CPU path's chi^2_1
    (tests/golden/lattice_316.npz, produced by the unmodified reference: 44.8 s per iteration there),
    Gauss-Newton keeps converging (chi^2 non-increasing, small 4th step)."""
    arr = lib.lattice_arrays(316)
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    chi2, ms = g.batch_resident(p, 4)
    stats = p.stats()
    assert stats["n_nodes"] == 99856 and stats["n_factors"] == 397531 and stats["not_spd"] == 0
    G = golden("lattice_316.npz")
    assert chi2[0] == pytest.approx(float(G["chi2"][0]), rel=1e-12) and chi2[0] == pytest.approx(23540091.696901, rel=1e-9)
    assert chi2[1] == pytest.approx(float(G["chi2"][1]), rel=CHI2_RTOL)  # reference CPU value after 1 iteration
    assert np.all(np.diff(chi2) <= 1e-9 * chi2[:-1])
    dx = g.deltas()
    assert np.max(np.abs(dx)) < 0.05                                     # Gauss-Newton is converging: 4th step is small
    p.destroy(); g.destroy()


def test_tutorial_incremental_mode_matches_reference_golden(lib):
    G = golden("tutorial_inc.npz")
    res = harness.run_tutorial(lib, batch_update_only=False)
    for k, (c, st) in enumerate(res):
        assert c == pytest.approx(float(G["chi2"][k]), rel=CHI2_RTOL, abs=1e-9)
        assert np.max(np.abs(st - G[f"states_{k}"])) < STATE_ATOL


@pytest.mark.parametrize("inc_fast", [1, 0])
def test_incremental_demo_matches_reference_schedule(lib, inc_fast):
    """Config 3 (first 650 poses of the M3500 demo, deterministic schedule): per-step chi^2 within 1e-6 of the
    reference's april_graph_cholesky_inc, IDENTICAL batch fall-back steps (232, 350, 508, 591 nodes), same
    final states.  Golden: tests/golden/m3500_inc_demo.npz, produced by the unmodified reference."""
    G = golden("m3500_inc_demo.npz")
    n = 650 if inc_fast else 360          # inc_fast=0: full re-plan per step (the slow, structure-agnostic path)
    lib.set_option("inc_fast", inc_fast)
    try:
        res = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=n, deterministic=True)
    finally:
        lib.set_option("inc_fast", 1)
    assert np.array_equal(res["was_batch"], G["was_batch"][:n])
    assert (np.nonzero(res["was_batch"])[0] + 1).tolist() == [k for k in (1, 232, 350, 508, 591) if k <= n]
    rel = np.abs(res["chi2"] - G["chi2"][:n]) / np.maximum(G["chi2"][:n], 1e-9)
    assert np.max(rel) < CHI2_RTOL, (int(np.argmax(rel)), float(np.max(rel)))


@pytest.mark.parametrize("opts", [
    {"inc_tail": 0},                                  # every step regenerates its fronts (one launch for the small ones)
    {"inc_tail": 0, "inc_one": 0},                    # ... as prologue + multi-level fronts + multi-level back substitution
    {"inc_multi": 0},                                 # ... one launch per level and direction (implies no k_inc_one / tail_refactor)
    {"inc_inline": 0},                                # patches read across PCIe instead of from the kernel arguments
    {"inc_one_spin": 0},                              # completion through hipStreamSynchronize
    {"inc_one_threads": 1024}, {"inc_one_threads": 256},
    {"inc_one_up": 1, "inc_one_dn": 1},               # k_inc_one for single-front steps only
    {"tail_poses": 8},                                # short tail fronts: new tail fronts open often
    {"inc_update": 0},                                # loop closures re-assemble and re-factorise their root paths (no low-rank updates)
    {"inc_update": 1, "inc_one_up": 16, "inc_one_dn": 16},   # low-rank updates inside k_inc_one (front after front in one workgroup) wherever the walk is short
], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_incremental_step_launch_forms_agree_with_the_reference_golden(lib, opts):
    """The launch forms of an incremental step (k_inc_one with tail_refactor, k_inc_one with regenerated fronts, multi-level
    launches, per-level launches; inline / PCIe patches; completion word / stream synchronise) on the first 420 poses of the
    demo: identical fall-back schedule, chi^2 within 1e-6 of the reference golden at every step."""
    G = golden("m3500_inc_demo.npz")
    n = 420
    with lib.options(**opts):
        res = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=n, deterministic=True)
    assert np.array_equal(res["was_batch"], G["was_batch"][:n])
    rel = np.abs(res["chi2"] - G["chi2"][:n]) / np.maximum(G["chi2"][:n], 1e-9)
    assert np.max(rel) < CHI2_RTOL, (int(np.argmax(rel)), float(np.max(rel)))


def test_loop_closures_take_low_rank_updates_and_match_the_golden(lib):
    """Round 4: the fronts on a loop closure's root path are UPDATED (front_update_body: three vectors per new factor travelling
    up the assembly tree), not re-assembled and re-factorised.  The first 700 poses of the demo: most fronts regenerated by
    loop-closure steps must have taken the update (stats.inc_fronts_updated), with the identical fall-back schedule and
    chi^2 within 1e-6 of the reference golden at every step; with the option off nothing is updated and the numbers agree."""
    G = golden("m3500_inc_demo.npz")
    n = 700
    out = {}
    for upd in (1, 0):
        lib.set_option("inc_update", upd)
        regen, updated = [], []

        def on_step(k, p, was_batch):
            if k and not was_batch:
                st = p.stats()
                if st["symbolic_reused"]:
                    regen.append(st["reserved0"]); updated.append(st["inc_fronts_updated"])
        try:
            res = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=n, deterministic=True, on_step=on_step)
        finally:
            lib.set_option("inc_update", 1)
        assert np.array_equal(res["was_batch"], G["was_batch"][:n])
        rel = np.abs(res["chi2"] - G["chi2"][:n]) / np.maximum(G["chi2"][:n], 1e-9)
        assert np.max(rel) < CHI2_RTOL, (upd, int(np.argmax(rel)), float(np.max(rel)))
        out[upd] = (np.array(regen), np.array(updated), res["chi2"])
    regen, updated, c1 = out[1]
    multi = regen >= 2                                    # steps that touched more than the last tail front
    assert multi.sum() > 100 and updated[multi].sum() >= 0.7 * (regen[multi] - 1).sum(), (int(multi.sum()), int(updated.sum()), int(regen[multi].sum()))
    assert np.all(updated <= np.maximum(regen - 1, 0))    # (the last tail front is always re-factorised)
    assert out[0][1].sum() == 0
    assert np.max(np.abs(c1 - out[0][2]) / np.maximum(out[0][2], 1e-9)) < 1e-9


def test_incremental_general_usage_falls_back_to_replanning(lib, oracle):
    """factors between two OLD poses and steps without a new pose do not fit the frozen structure of the fast
    path: the library must notice and re-plan; results = exact solve of the incremental system on all poses
    (naffected > 5 here, so the reference updates every pose too)"""
    arr = datasets.random_pose_graph(60, 30, 8)
    st, fa, fb, z, W = arr
    keep = np.ones(len(fa), bool)
    late = [k for k in range(len(fa)) if fb[k] >= 0 and abs(int(fa[k]) - int(fb[k])) > 1][-6:]
    keep[late] = False
    g = lib.new_graph(); g.build_from_arrays(st, fa[keep], fb[keep], z[keep], W[keep]); p = lib.new_param(nthreshold=10**6)
    g.cholesky(p)
    lp = g.l_points()
    for k in late:                       # six loop closures between existing poses, one call each, no new pose
        g.add_factor_xyt(int(fa[k]), int(fb[k]), z[k], W[k])
    p.c.batch_time = 1e300
    g.cholesky_inc(p)
    order = np.concatenate([np.nonzero(keep)[0], late])
    lam = np.full(len(st), 1e-4)
    dx = oracle.solve_system(lp, lp, fa[order], fb[order], z[order], W[order], lam)
    pred = lp + dx; pred[:, 2] = [oracle.mod2pi(v) for v in pred[:, 2]]
    assert np.max(np.abs(g.states() - pred)) < 1e-8
    p.destroy(); g.destroy()


def test_lattice_200k_big_path_is_race_free(lib):
    """K = 450 (202 500 poses): hundreds of multi-workgroup fronts per level.  Guards the inter-workgroup hazard the
    panel step once had (row tiles re-reading a diagonal block another workgroup was overwriting): the step must
    factor (no bad pivot) and Gauss-Newton must make progress, twice in a row with identical results."""
    runs = []
    for _ in range(2):
        g = lib.new_graph(); lib.dll.aprilsam_amd_make_lattice(g.ptr, 450); p = lib.new_param()
        chi2, ms = g.batch_resident(p, 2)
        assert p.stats()["not_spd"] == 0
        assert chi2[1] < 0.03 * chi2[0] and chi2[2] <= chi2[1]
        runs.append(chi2)
        p.destroy(); g.destroy()
    assert np.array_equal(runs[0], runs[1])


def test_lattice_1m_single_gpu_matches_the_recorded_trace(lib):
    """config 5 on one GPU (10^6 poses, 27 GB of fronts): chi^2 at the start, after 1 and after 3 Gauss-Newton iterations
    equal the values the sharded runs are checked against (bench.LATTICE1M_CHI2; SURVEY 8(d): parity at 1 M is against
    the build's own single-GPU path, itself reference-checked up to 100 k)"""
    import bench
    g = lib.new_graph(); lib.dll.aprilsam_amd_make_lattice(g.ptr, 1000); p = lib.new_param()
    chi2, _ = g.batch_resident(p, 3)
    assert p.stats()["not_spd"] == 0
    got = np.array([chi2[0], chi2[1], chi2[3]]); want = np.array(bench.LATTICE1M_CHI2)
    assert np.max(np.abs(got - want) / want) < 1e-9, (got, want)
    p.destroy(); g.destroy()


def test_many_tiny_and_odd_graphs_agree_with_oracle(lib, oracle):
    """2 .. 45 poses, from a bare chain to nearly complete graphs, several leaf sizes: fronts of one pose, panels narrower
    than an MFMA k-step, children with a single block, empty update blocks"""
    rng = np.random.default_rng(2024)
    worst = 0.0
    try:
        for trial in range(60):
            n = int(rng.integers(2, 46))
            extra = int(rng.integers(0, max(1, n * (n - 1) // 2 - (n - 1)) // (1 if trial % 3 == 0 else 4) + 1))
            arr = datasets.random_pose_graph(n, extra, 1000 + trial, spread=4.0)
            lib.set_option("leaf_nodes", [1, 2, 5, 16][trial % 4])
            oc, ost = oracle.iterate(arr, 2)
            chi2, snaps, stats = run_batch(lib, arr, 2)
            assert stats["not_spd"] == 0
            assert np.max(np.abs(chi2 - oc)) < 1e-8 * max(oc[0], 1e-9), (trial, n, extra, chi2, oc)
            worst = max(worst, float(np.max(np.abs(snaps[-1][0] - ost))))
    finally:
        lib.set_option("leaf_nodes", 16)
    assert worst < STATE_ATOL


def _random_growth(lib, rng_seed, steps, nthreshold, old_old=False, observe_every=1):
    """grow a pose graph step by step through the reference API: every step adds 1-2 poses with odometry, often loop
    closures to random old poses (full information matrices); old_old: sometimes also a factor between two old poses"""
    rng = np.random.default_rng(rng_seed)
    g = lib.new_graph(); p = lib.new_param(nthreshold=nthreshold, delta_xy=0.05, delta_theta=0.05)
    truth = [np.zeros(3)]
    g.add_node_xyt(truth[0]); g.add_factor_xytpos(0, [0, 0, 0], datasets.PRIOR_W)
    g.cholesky(p)
    trace = []

    def rel(a, b):
        c, s = np.cos(a[2]), np.sin(a[2]); dx, dy = b[0] - a[0], b[1] - a[1]
        return np.array([c * dx + s * dy, -s * dx + c * dy, b[2] - a[2]])

    def info():
        M = rng.normal(size=(3, 3)); Wk = M @ M.T + np.diag([40.0, 40.0, 120.0]); return (Wk + Wk.T) / 2

    for step in range(steps):
        for _ in range(1 + (step % 7 == 3)):
            last = truth[-1]
            new = np.array([last[0] + np.cos(last[2]) * 0.8, last[1] + np.sin(last[2]) * 0.8, last[2] + rng.uniform(-0.6, 0.6)])
            truth.append(new); n = len(truth) - 1
            g.add_node_xyt(new + rng.normal(0, [0.15, 0.15, 0.04]))
            g.add_factor_xyt(n - 1, n, rel(truth[n - 1], new) + rng.normal(0, [0.03, 0.03, 0.01]), info())
        n = len(truth) - 1
        if n > 4 and rng.random() < 0.6:
            for _ in range(int(rng.integers(1, 3))):
                o = int(rng.integers(0, n - 1))
                a, b = (o, n) if rng.random() < 0.5 else (n, o)
                g.add_factor_xyt(a, b, rel(truth[a], truth[b]) + rng.normal(0, [0.03, 0.03, 0.01]), info())
        if old_old and n > 10 and step % 11 == 5:
            a, b = sorted(rng.choice(n - 1, 2, replace=False).tolist())
            g.add_factor_xyt(a, b, rel(truth[a], truth[b]) + rng.normal(0, [0.03, 0.03, 0.01]), info())
        p.c.batch_time = 1e300
        g.cholesky_inc(p)
        if step % observe_every == observe_every - 1 or step == steps - 1:
            trace.append((g.chi2(), g.states()))        # (april_graph_chi2 reads every node: it also brings the library's state mirrors in step)
    p.destroy(); g.destroy()
    return trace


def _recent_pose_growth(lib, rng_seed, steps, nthreshold, tail_poses):
    """growth that lives near the newest poses -- what tail_refactor serves: every step adds 1-3 poses with odometry, often a
    factor between two of the last 6 poses (either orientation), sometimes a prior on a recent pose or a second factor on
    the same pair, now and then a loop closure to an old pose (the general path in between)"""
    rng = np.random.default_rng(rng_seed)
    lib.set_option("tail_poses", tail_poses)
    try:
        g = lib.new_graph(); p = lib.new_param(nthreshold=nthreshold, delta_xy=0.05, delta_theta=0.05)
        truth = [np.zeros(3)]
        g.add_node_xyt(truth[0]); g.add_factor_xytpos(0, [0, 0, 0], datasets.PRIOR_W)
        g.cholesky(p)
        trace = []

        def rel(a, b):
            c, s = np.cos(a[2]), np.sin(a[2]); dx, dy = b[0] - a[0], b[1] - a[1]
            return np.array([c * dx + s * dy, -s * dx + c * dy, b[2] - a[2]])

        def info():
            M = rng.normal(size=(3, 3)); Wk = M @ M.T + np.diag([40.0, 40.0, 120.0]); return (Wk + Wk.T) / 2

        for step in range(steps):
            n0 = len(truth)                                   # first pose of this call
            for _ in range(int(rng.integers(1, 4))):
                last = truth[-1]
                new = np.array([last[0] + np.cos(last[2]) * 0.8, last[1] + np.sin(last[2]) * 0.8, last[2] + rng.uniform(-0.6, 0.6)])
                truth.append(new); n = len(truth) - 1
                g.add_node_xyt(new + rng.normal(0, [0.15, 0.15, 0.04]))
                g.add_factor_xyt(n - 1, n, rel(truth[n - 1], new) + rng.normal(0, [0.03, 0.03, 0.01]), info())
            n = len(truth) - 1
            if n > 7 and rng.random() < 0.7:
                for _ in range(int(rng.integers(1, 4))):      # one end is a pose of THIS call (factors between two older poses: see
                    b = int(rng.integers(n0, n + 1))          # test_factors_between_two_old_poses_against_the_live_reference)
                    a = int(rng.choice([q for q in range(n - 6, n + 1) if q != b]))
                    if rng.random() < 0.5:
                        a, b = b, a
                    g.add_factor_xyt(a, b, rel(truth[a], truth[b]) + rng.normal(0, [0.03, 0.03, 0.01]), info())
            if n > 3 and rng.random() < 0.25:
                o = int(rng.integers(max(0, n - 5), n + 1))
                g.add_factor_xytpos(o, truth[o] + rng.normal(0, [0.05, 0.05, 0.02]), np.diag([25.0, 25.0, 40.0]))
            if n > 12 and rng.random() < 0.12:
                o = int(rng.integers(0, n - 8))
                g.add_factor_xyt(o, n, rel(truth[o], truth[n]) + rng.normal(0, [0.03, 0.03, 0.01]), info())
            p.c.batch_time = 1e300
            g.cholesky_inc(p)
            trace.append((g.chi2(), g.states()))
        p.destroy(); g.destroy()
        return trace
    finally:
        lib.set_option("tail_poses", 28)


@pytest.mark.parametrize("seed,nthreshold,tail_poses", [(11, 10 ** 6, 28), (12, 40, 28), (13, 10 ** 6, 9), (14, 25, 16)])
def test_growth_among_the_newest_poses_matches_the_live_reference(lib, reflib, seed, nthreshold, tail_poses):
    """tail_refactor's territory (several new poses per call, factors among the last few poses in both orientations, priors,
    repeated pairs) interleaved with loop closures, with short tail fronts too (a tail front fills up and the next one opens
    every few steps): chi^2 and every state against the unmodified reference, step by step."""
    ours = _recent_pose_growth(lib, seed, 120, nthreshold, tail_poses)
    ref = _recent_pose_growth_ref(reflib, seed, 120, nthreshold)
    for k, ((c1, s1), (c2, s2)) in enumerate(zip(ours, ref)):
        assert abs(c1 - c2) <= 1e-6 * max(c2, 1.0), (k, c1, c2)
        assert np.max(np.abs(s1 - s2)) < 1e-6, k


def _recent_pose_growth_ref(reflib, seed, steps, nthreshold):
    class _NoOptions:            # the reference library has no options: same driver, set_option ignored
        def __init__(self, L): self.L = L
        def set_option(self, *a): pass
        def __getattr__(self, k): return getattr(self.L, k)
    return _recent_pose_growth(_NoOptions(reflib), seed, steps, nthreshold, 28)


@pytest.mark.parametrize("seed,nthreshold", [(4, 10 ** 6), (5, 30)])
def test_incremental_calls_back_to_back_without_a_chi2_call_in_between(lib, reflib, seed, nthreshold):
    """A caller that does not evaluate chi^2 between incremental calls: the library's pinned state mirrors and device copies
    are then kept in step by the library alone (apply_visits moves the mirrors, the next call patches the device copies of
    the poses it updated -- or loads every state after a full walk).  States compared with the live reference every 25 steps."""
    ours = _random_growth(lib, seed, 150, nthreshold, observe_every=25)
    ref = _random_growth(reflib, seed, 150, nthreshold, observe_every=25)
    assert len(ours) == len(ref) == 6
    for k, ((c1, s1), (c2, s2)) in enumerate(zip(ours, ref)):
        assert abs(c1 - c2) <= 1e-6 * max(c2, 1.0), (k, c1, c2)
        assert np.max(np.abs(s1 - s2)) < 1e-6, k


@pytest.mark.parametrize("extend", [1, 0])
def test_batch_calls_on_a_growing_graph_match_the_live_reference(lib, reflib, extend):
    """the reference demo's --batch_update_only mode (examples/aprilsam_demo.c:224-228): one april_graph_cholesky per new
    pose on a graph that grows by a pose and its factors each time.  With batch_extend (default) the library keeps the
    plan and appends tail fronts instead of re-planning per call; both ways every step's chi^2 and the final states must
    equal the unmodified reference's."""
    from aprilsam_amd import harness
    n = 420
    lib.set_option("batch_extend", extend)
    try:
        ours = harness.run_demo(lib, datasets.m3500_arrays(), batch_update_only=True, max_poses=n)
    finally:
        lib.set_option("batch_extend", 1)
    ref = harness.run_demo(reflib, datasets.m3500_arrays(), batch_update_only=True, max_poses=n)
    assert np.max(np.abs(ours["chi2"] - ref["chi2"]) / np.maximum(ref["chi2"], 1e-6)) < 1e-6
    assert np.max(np.abs(ours["final_states"] - ref["final_states"])) < 1e-6
    if extend:      # ... and it is what makes the cold call cheap: far less time than re-planning every step
        assert ours["ms"][50:].mean() < 2.0


def test_factors_edited_in_place_are_seen_by_the_next_call(lib, reflib):
    """the reference re-reads every factor object on every call (aprilsam.c:152-190, april_graph.c:79-98): a caller may
    edit z / W of an existing factor in place, or replace a factor object, between two calls.  Default options
    (trust_factor_cache = 0) must follow; both libraries are driven by the same code and compared call by call."""
    arr = datasets.random_pose_graph(120, 90, 17)
    out = []
    for L in (lib, reflib):
        g = L.new_graph(); g.build_from_arrays(*arr); p = L.new_param()
        tr = []
        g.cholesky(p); tr.append((g.chi2(), g.states()))
        f = g.factor(37)                                   # measurement and information edited in place
        f.u.z[0] += 0.25; f.u.z[2] -= 0.1
        for k in (0, 4, 8):
            f.u.W.contents.data[k] *= 3.0
        tr.append((g.chi2(), g.states()))                  # chi^2 sees the edit at once
        g.cholesky(p); tr.append((g.chi2(), g.states()))
        g.factor(5).u.z[1] += 0.4                          # ... and again on a warm (plan-cached) call
        g.cholesky(p); tr.append((g.chi2(), g.states()))
        out.append(tr)
        p.destroy(); g.destroy()
    for k, ((c1, s1), (c2, s2)) in enumerate(zip(*out)):
        assert abs(c1 - c2) <= 1e-9 * max(c2, 1.0), (k, c1, c2)
        assert np.max(np.abs(s1 - s2)) < 1e-8, k
    assert abs(out[0][1][0] - out[0][0][0]) > 1e-3          # the edit did change chi^2


def test_warm_call_speculates_on_unchanged_factors_and_starts_over_after_an_edit(lib, reflib):
    """batch_impl: a warm call launches on the packed factor copies and reads the factor objects under the GPU's work
    (option speculate_factors).  An unchanged graph takes that path (stats.reserved1 = 0, one run); an edit in place voids
    the run and the call starts over (reserved1 = 1) -- and the states equal the reference's either way, with the option
    off as well."""
    arr = datasets.random_pose_graph(150, 110, 23)
    res = {}
    for name, L, spec in (("spec", lib, 1), ("nospec", lib, 0), ("ref", reflib, None)):
        if spec is not None:
            L.set_option("speculate_factors", spec)
        try:
            g = L.new_graph(); g.build_from_arrays(*arr); p = L.new_param()
            tr, flags = [], []
            for call in range(5):
                if call == 2:
                    g.factor(11).u.z[0] += 0.3                 # edited in place before the third call
                if call == 4:
                    g.factor(40).u.W.contents.data[4] *= 2.0   # ... and the information before the fifth
                g.cholesky(p)
                tr.append((g.chi2(), g.states()))
                if spec is not None:
                    flags.append(p.stats()["reserved1"])
            res[name] = (tr, flags)
            p.destroy(); g.destroy()
        finally:
            if spec is not None:
                L.set_option("speculate_factors", 1)
    assert res["spec"][1] == [0, 0, 1, 0, 1], res["spec"][1]      # cold, warm speculative, voided, speculative, voided
    assert res["nospec"][1] == [0, 0, 0, 0, 0]
    for name in ("spec", "nospec"):
        for k, ((c1, s1), (c2, s2)) in enumerate(zip(res[name][0], res["ref"][0])):
            assert abs(c1 - c2) <= 1e-9 * max(c2, 1.0), (name, k, c1, c2)
            assert np.max(np.abs(s1 - s2)) < 1e-8, (name, k)


def _growth_with_late_priors(lib, steps=60):
    """incremental growth where xytpos priors arrive in the middle of the run (the reference evaluates a prior at the
    node's STATE of that call, april_graph_xytpos.c:83-85, not at its l_point) and a batch fall-back happens later"""
    rng = np.random.default_rng(5)
    g = lib.new_graph(); p = lib.new_param(nthreshold=18, delta_xy=0.05, delta_theta=0.05)
    truth = [np.zeros(3)]
    g.add_node_xyt(truth[0]); g.add_factor_xytpos(0, [0, 0, 0], datasets.PRIOR_W)
    g.cholesky(p)
    trace = []
    W = np.diag([60.0, 60.0, 150.0])
    for step in range(steps):
        last = truth[-1]
        new = np.array([last[0] + np.cos(last[2]) * 0.9, last[1] + np.sin(last[2]) * 0.9, last[2] + rng.uniform(-0.5, 0.5)])
        truth.append(new); n = len(truth) - 1
        g.add_node_xyt(new + rng.normal(0, [0.2, 0.2, 0.05]))
        c, s_ = np.cos(last[2]), np.sin(last[2]); dx, dy = new[0] - last[0], new[1] - last[1]
        g.add_factor_xyt(n - 1, n, np.array([c * dx + s_ * dy, -s_ * dx + c * dy, new[2] - last[2]]) + rng.normal(0, [0.03, 0.03, 0.01]), W)
        if step % 9 == 4:                                   # a GPS-like prior on the new pose and one on an older pose
            g.add_factor_xytpos(n, new + rng.normal(0, [0.05, 0.05, 0.02]), np.diag([25.0, 25.0, 40.0]))
            o = int(rng.integers(0, n))
            g.add_factor_xytpos(o, truth[o] + rng.normal(0, [0.05, 0.05, 0.02]), np.diag([25.0, 25.0, 40.0]))
        p.c.batch_time = 1e300
        g.cholesky_inc(p)
        trace.append((g.chi2(), g.states()))
    p.destroy(); g.destroy()
    return trace


def test_priors_added_between_incremental_calls_match_the_live_reference(lib, reflib):
    ours = _growth_with_late_priors(lib)
    ref = _growth_with_late_priors(reflib)
    for k, ((c1, s1), (c2, s2)) in enumerate(zip(ours, ref)):
        assert abs(c1 - c2) <= 1e-6 * max(c2, 1.0), (k, c1, c2)
        assert np.max(np.abs(s1 - s2)) < 1e-6, k


@pytest.mark.parametrize("seed,nthreshold", [(1, 25), (2, 10 ** 6), (3, 8)])
def test_random_incremental_growth_matches_the_live_reference(lib, reflib, seed, nthreshold):
    """irregular incremental use (several poses per call, loop closures in both orientations, frequent or no batch
    fall-backs) step by step against the unmodified reference: chi^2 and every state.
    Factors between two OLD poses are left out on purpose: when the two poses sit in different branches of the
    reference's elimination tree, its partial re-factorisation (aprilsam.c:850-906, children first over the OLD tree)
    finalises one row before the other has updated it, and its result is no longer the solution of its own system
    (measured: 3e-2 off the exact solve, this build 1e-4 incl. the prior's relinearisation) -- see
    test_incremental_general_usage_falls_back_to_replanning for that case against the exact solve."""
    ours = _random_growth(lib, seed, 140, nthreshold)
    ref = _random_growth(reflib, seed, 140, nthreshold)
    for k, ((c1, s1), (c2, s2)) in enumerate(zip(ours, ref)):
        assert abs(c1 - c2) <= 1e-6 * max(c2, 1.0), (k, c1, c2)
        assert np.max(np.abs(s1 - s2)) < 1e-6, k
