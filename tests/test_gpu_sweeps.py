"""Bounded slices (about ten cases, under 20 s each) of the randomised / structured sweeps of tools/stress_*.py (tests/support/sweeps.py):
random loop-closure graphs whose dense root fronts take every width modulo the 32- and 128-column blockings, chains / stars / complete /
banded / comb graphs, odd lattices, random incremental growth, growth near the newest poses, one batch call per step on a growing graph --
against the oracle and the live reference.  The random-graph slice runs a second time with option pool_guard: NaN-filled guard bands
behind every frontal array, so a read past a front that is used changes the result instead of depending on where the allocation ends
(round 4's bug faulted only when the front was the last array of the pool), and a write past a front is reported (error -16)."""
import numpy as np
import pytest

import tests.test_gpu_parity as T
from tests.support import sweeps

pytestmark = pytest.mark.gpu
BIG = (dict(), dict(small_lds_kb=0), dict(small_lds_kb=0, blk_backsolve=0), dict(small_lds_kb=0, syrk_pair_tiles=1))


@pytest.mark.parametrize("guard", [0, 512])
def test_random_loop_closure_graphs(lib, oracle, guard):
    with lib.options(pool_guard=guard):
        sweeps.sweep_batch(lib, oracle, sweeps.random_graph_cases(7, 6, 300, 2200) + sweeps.random_graph_cases(14, 4, 300, 1200), BIG, 1e-7, 1e-5, log=lambda s: None)


def test_a_write_into_a_guard_band_is_reported(lib):
    """the guard check itself: with pool_guard on, a healthy step reports nothing (above); here the check is pointed at a band that a
    kernel legitimately writes -- guard bands shorter than the alignment padding cannot exist, so instead the option is switched on
    AFTER the plan was made: the bands it then declares lie inside live frontal arrays and must be reported as overwritten"""
    arr = sweeps.random_graph_cases(7, 1, 300, 400)[0][1]
    with lib.options(pool_guard=64):
        g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
        g.cholesky(p)
        assert p.stats()["error_code"] == 0
        assert lib.dll.aprilsam_amd_debug_guard_selftest(p.ptr) == -16         # declares a band inside the first front, runs the check
        lib.clear_error()
        p.destroy(); g.destroy()


def test_structured_graphs(lib, oracle):
    cases = [("chain", 1500), ("chain", 97), ("star", 700), ("star", 130), ("complete", 60), ("complete", 140), ("two", 900), ("band", 1100), ("band", 257), ("comb", 1300)]
    sweeps.sweep_batch(lib, oracle, [(f"{k} n={n}", sweeps.structured(k, n, 300 + i)) for i, (k, n) in enumerate(cases)],
                       (dict(), dict(small_lds_kb=0), dict(small_lds_kb=16, leaf_nodes=6)), 1e-6, 1e-5, log=lambda s: None)


def test_soak_slice_every_result_bitwise_equal_to_the_first_of_its_case(lib):
    """a few seconds of tools/soak_batch.py: graphs in random order, fresh graph + param each time, results compared bitwise with the first
    run of the same case and options.  (What it guards -- a front of a multi-level launch consuming a child's block of the iteration
    before -- showed up once in about 10^4 solves before the flags carried the iteration number: the hour-long version is the tool.)"""
    lines = []
    runs, bad = sweeps.soak(lib, 4.0, 11, log=lines.append)
    assert runs > 50 and bad == 0, lines


def test_poisoned_hand_overs_never_reach_a_result(lib, oracle):
    """Debug option pool_poison: before every step the update block of every front (what a parent's extend-add reads) and x (what a child's
    back substitution gathers) are NaN-filled.  A dependency wait of a multi-level launch that passes early -- round 5's release defect,
    about one solve in 10^4 on chain-like graphs, invisible to every test because the stale numbers were the previous run's correct ones --
    then produces NaN or "not positive definite" with certainty.  The 14 soak cases x 3 option sets x 3 rounds, 16 calls each (a cold call,
    then the captured graph on states that keep changing): 2 016 solves, none may show a NaN, all bitwise equal to the first of their case;
    and the numbers are the unpoisoned ones (oracle, two iterations)."""
    lines = []
    solves, bad = sweeps.poison_soak(lib, 3, 16, 5, log=lines.append)
    assert solves >= 2000 and bad == 0, lines
    cases = sweeps.soak_cases()
    with lib.options(pool_poison=1):
        sweeps.sweep_batch(lib, oracle, [cases[0], cases[7], cases[9]], (dict(), dict(small_lds_kb=0)), 1e-6, 1e-5, log=lambda s: None)


def test_the_poison_is_seen_when_a_wait_is_skipped(lib):
    """... and the negative control: the poison does reach the result when a hand-over IS read early.  persist = 1 with the dependency
    waits of the multi-level launches switched off (debug option skip_flag_waits): a chain-like graph must come back NaN / not positive
    definite / with a dependency time-out -- if it does not, pool_poison poisons nothing the launches read."""
    arr = sweeps.soak_cases()[0][1]
    seen = 0
    with lib.options(pool_poison=1, skip_flag_waits=1):
        for _ in range(3):
            g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
            g.cholesky(p); s = p.stats(); st = g.states(); c = g.chi2()
            seen += int(s["not_spd"] != 0 or s["error_code"] != 0 or not np.all(np.isfinite(st)) or not np.isfinite(c))
            lib.clear_error(); p.destroy(); g.destroy()
    assert seen >= 1


@pytest.mark.parametrize("opts", [dict(), dict(inc_one=0, inc_tail=0), dict(inc_update=0, inc_one=0)], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()) or "default")
def test_incremental_demo_under_pool_poison(lib, opts):
    """the incremental path's multi-level launches (regenerated fronts, low-rank updates, the partial back substitution) with everything
    they hand over poisoned first: the first 700 poses of the M3500 demo keep the reference's fall-back schedule and chi^2 trace"""
    import os
    from aprilsam_amd import datasets, harness
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m3500_inc_demo.npz"))
    n = 700
    with lib.options(pool_poison=1, **opts):
        res = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=n, deterministic=True)
    assert np.all(np.isfinite(res["chi2"])) and np.all(np.isfinite(res["final_states"]))
    assert np.array_equal(res["was_batch"], G["was_batch"][:n])
    rel = np.abs(res["chi2"] - G["chi2"][:n]) / np.maximum(G["chi2"][:n], 1e-9)
    assert np.max(rel) < 1e-6, (int(np.argmax(rel)), float(np.max(rel)))


def test_odd_lattices(lib, oracle):
    sweeps.sweep_batch(lib, oracle, [(f"lattice K={K}", lib.lattice_arrays(K)) for K in (37, 91)],
                       (dict(), dict(small_lds_kb=0), dict(small_lds_kb=48, leaf_nodes=24), dict(small_lds_kb=0, leaf_nodes=7, syrk_xcd_order=1, syrk_small_tiles=1 << 30),
                        dict(small_lds_kb=0, syrk_pair_tiles=1, syrk_group=2), dict(small_lds_kb=0, syrk_pair_tiles=1, syrk_group=4)),
                       1e-8, 1e-6, log=lambda s: None)


@pytest.mark.parametrize("seed", range(100, 106))
def test_random_incremental_growth_against_the_live_reference(lib, reflib, seed):
    nth = [10 ** 6, 40, 12, 25][seed % 4]
    ec, es = sweeps.compare_traces(T._random_growth(lib, seed, 120, nth), T._random_growth(reflib, seed, 120, nth))
    assert ec < 1e-6 and es < 1e-6, (ec, es)


@pytest.mark.parametrize("seed,tp", [(200, 28), (201, 9), (202, 16), (203, 12)])
def test_growth_near_the_newest_poses_against_the_live_reference(lib, reflib, seed, tp):
    nth = [10 ** 6, 30][seed % 2]
    ec, es = sweeps.compare_traces(T._recent_pose_growth(lib, seed, 110, nth, tp), T._recent_pose_growth_ref(reflib, seed, 110, nth))
    assert ec < 1e-6 and es < 1e-6, (ec, es)


@pytest.mark.parametrize("seed,ext", [(300, 1), (300, 0), (301, 1), (301, 0)])
def test_one_batch_call_per_step_on_a_growing_graph(lib, reflib, seed, ext):
    G = type(lib.new_graph())
    orig = G.cholesky_inc
    G.cholesky_inc = lambda self, p: self.cholesky(p)          # the same growth, one BATCH call per step (the demo's --batch_update_only)
    try:
        with lib.options(batch_extend=ext):
            ours = T._random_growth(lib, seed, 110, 10 ** 6, old_old=(seed % 2 == 0))
        ref = T._random_growth(reflib, seed, 110, 10 ** 6, old_old=(seed % 2 == 0))
    finally:
        G.cholesky_inc = orig
    ec, es = sweeps.compare_traces(ours, ref)
    assert ec < 1e-6 and es < 1e-6, (ec, es)
