"""One process, several device slots (aprilsam_amd_param_set_device): params bound to different slots are solved CONCURRENTLY from
different threads -- each slot has its own lock, every call makes its slot's device current on the calling thread, nothing on the
path is process-global (the reference's solver has no globals either: SURVEY.md section 8(b) "Threading").  On this one-GPU box the
slots share device 0 (slot s runs on device s % device_count), which is exactly the test: what would break is shared state, not the
device.  A node with N devices runs the same code with slot s on device s -- no fork, no RCCL bootstrap for independent solves."""
import threading

import numpy as np
import pytest

from aprilsam_amd import datasets, harness
from tests.conftest import golden

pytestmark = pytest.mark.gpu


def _batch_worker(lib, slot, arr, iters, out, barrier):
    try:
        g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
        assert lib.dll.aprilsam_amd_param_set_device(p.ptr, slot) == 0
        barrier.wait()
        chi2 = [g.chi2()]
        for _ in range(iters):
            g.cholesky(p); chi2.append(g.chi2())
        out[slot] = (np.array(chi2), g.states(), p.stats(), lib.dll.aprilsam_amd_param_get_device(p.ptr))
        p.destroy(); g.destroy()
    except BaseException as e:          # noqa: BLE001 -- reported by the main thread
        out[slot] = e


def test_two_slots_solve_concurrently_from_two_threads(lib, oracle):
    arrs = {0: datasets.random_pose_graph(1500, 900, 3), 1: datasets.random_pose_graph(1200, 1000, 41)}
    want = {s: oracle.iterate(a, 3) for s, a in arrs.items()}
    for rep in range(3):                                    # (three rounds: different interleavings)
        out = {}; barrier = threading.Barrier(2)
        th = [threading.Thread(target=_batch_worker, args=(lib, s, arrs[s], 3, out, barrier)) for s in (0, 1)]
        [t.start() for t in th]; [t.join(120) for t in th]
        for s in (0, 1):
            assert not isinstance(out.get(s), BaseException), out.get(s)
            chi2, st, stats, dev = out[s]
            assert stats["error_code"] == 0 and stats["not_spd"] == 0
            assert dev == s % lib.device_count()
            assert np.max(np.abs(chi2 - want[s][0]) / want[s][0]) < 1e-8
            assert np.max(np.abs(st - want[s][1])) < 1e-6


def test_an_incremental_run_and_a_batch_loop_side_by_side(lib, oracle):
    """slot 2: the first 300 poses of the M3500 demo through april_graph_cholesky_inc; slot 5: batch iterations of another graph; slot 7:
    the 100 x 100 lattice -- three threads, three slots, golden / oracle results each"""
    G = golden("m3500_inc_demo.npz")
    n = 300
    out = {}

    def inc_worker():
        try:
            state = {}

            def bind(k, p, wb):
                pass
            # (run_demo creates its own param: bind it through a wrapper of new_param)
            orig = lib.new_param

            def new_param(**kw):
                p = orig(**kw)
                if threading.current_thread().name == "inc":
                    assert lib.dll.aprilsam_amd_param_set_device(p.ptr, 2) == 0
                return p
            lib.new_param = new_param
            try:
                out["inc"] = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=n, deterministic=True, on_step=bind)
            finally:
                lib.new_param = orig
            del state
        except BaseException as e:      # noqa: BLE001
            out["inc"] = e
    arr = datasets.random_pose_graph(900, 700, 8)
    lat = lib.lattice_arrays(100)
    want = oracle.iterate(arr, 4)
    want_lat = oracle.iterate(lat, 1)
    barrier = threading.Barrier(2)
    th = [threading.Thread(target=inc_worker, name="inc"), threading.Thread(target=_batch_worker, args=(lib, 5, arr, 4, out, barrier)),
          threading.Thread(target=_batch_worker, args=(lib, 7, lat, 1, out, barrier))]
    [t.start() for t in th]; [t.join(300) for t in th]
    for k in ("inc", 5, 7):
        assert not isinstance(out.get(k), BaseException), (k, out.get(k))
    res = out["inc"]
    assert np.array_equal(res["was_batch"], G["was_batch"][:n])
    rel = np.abs(res["chi2"] - G["chi2"][:n]) / np.maximum(G["chi2"][:n], 1e-9)
    assert rel.max() < 1e-6
    for k, w in ((5, want), (7, want_lat)):
        chi2, st, stats, dev = out[k]
        assert stats["error_code"] == 0 and np.max(np.abs(chi2 - w[0]) / w[0]) < 1e-8 and np.max(np.abs(st - w[1])) < 1e-6


def test_a_param_moved_to_another_slot_starts_from_scratch_and_bad_slots_are_refused(lib, oracle):
    arr = datasets.random_pose_graph(300, 200, 3)
    oc, ost = oracle.iterate(arr, 2)
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    g.cholesky(p)
    assert lib.dll.aprilsam_amd_param_set_device(p.ptr, 3) == 0          # drops the plan; the graph's device copies follow at the next call
    g.cholesky(p)
    assert p.stats()["symbolic_reused"] == 0
    assert abs(g.chi2() - oc[2]) < 1e-8 * oc[2] and np.max(np.abs(g.states() - ost)) < 1e-6
    assert lib.dll.aprilsam_amd_param_set_device(p.ptr, -1) == -1 and lib.dll.aprilsam_amd_param_set_device(p.ptr, 64) == -1
    assert lib.dll.aprilsam_amd_param_set_device(p.ptr, 5) == 0
    lib.dll.april_graph_cholesky_param_init(p.ptr)                   # a re-initialised param is back on the default slot
    assert lib.dll.aprilsam_amd_param_get_device(p.ptr) == 0
    g.cholesky(p)                                                    # and still solves there
    assert np.isfinite(g.chi2()) and g.chi2() <= oc[2] * (1 + 1e-9)
    p.destroy(); g.destroy()
