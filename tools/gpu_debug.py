#!/usr/bin/env python3
"""GPU bring-up script: runs a ladder of cases through the C-ABI and prints the error vs the oracle for each,
without stopping at the first failure.  Usage (GPU box): python tools/gpu_debug.py [--quick]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from aprilsam_amd import datasets, host  # noqa: E402
from tests.support.oracle_binding import Oracle  # noqa: E402

lib = host.SolverLib()
orc = Oracle()
print("version:", lib.version(), "devices:", lib.device_count(), flush=True)


def case(name, arr, iters=1, opts=None, tol=1e-6):
    opts = opts or {}
    for k, v in opts.items():
        lib.set_option(k, v)
    try:
        g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
        c = [g.chi2()]
        t0 = time.perf_counter()
        for _ in range(iters):
            g.cholesky(p); c.append(g.chi2())
        ms = (time.perf_counter() - t0) * 1e3 / iters
        st = g.states()
        stats = p.stats()
        oc, ost = orc.iterate(arr, iters)
        err_c = np.max(np.abs(np.array(c) - oc) / np.maximum(np.abs(oc), 1e-300))
        err_s = np.max(np.abs(st - ost))
        ok = err_c < tol and err_s < 1e-5
        print(f"[{'OK ' if ok else 'BAD'}] {name:34s} N={len(arr[0]):6d} F={len(arr[1]):7d} fronts={stats['n_fronts']:5d} lev={stats['n_levels']:2d} "
              f"maxm={stats['max_front_rows']:4d} chi2_relerr={err_c:.2e} state_err={err_s:.2e} notspd={stats['not_spd']} ms/iter={ms:.2f}", flush=True)
        if not ok:
            print("      chi2 gpu   :", c[:4], "\n      chi2 oracle:", oc[:4].tolist(), flush=True)
        p.destroy(); g.destroy()
        return ok
    except Exception:
        traceback.print_exc()
        return False
    finally:
        for k in opts:
            lib.set_option(k, {"small_lds_kb": 156, "leaf_nodes": 16, "use_graph": 1}[k])


quick = "--quick" in sys.argv
res = []
tiny = (np.array([[0.1, 0.2, 0.3], [1.0, 0.1, 0.2]]), np.array([0, 0], np.int32), np.array([-1, 1], np.int32),
        np.array([[0, 0, 0], [1, 0, 0.0]]), np.vstack([datasets.PRIOR_W, np.diag([10.0, 10.0, 5.0]).reshape(9)]))
res.append(case("tiny 2 nodes (1 front)", tiny, 2, {"use_graph": 0}))
res.append(case("random 12 (leaf 4)", datasets.random_pose_graph(12, 6, 0), 3, {"leaf_nodes": 4, "use_graph": 0}))
res.append(case("random 80 small path", datasets.random_pose_graph(80, 60, 1), 3, {"use_graph": 0}))
res.append(case("random 80 BIG path only", datasets.random_pose_graph(80, 60, 1), 3, {"small_lds_kb": 0, "use_graph": 0}))
res.append(case("random 80 MEDIUM path only", datasets.random_pose_graph(80, 60, 1), 3, {"small_lds_kb": 0, "use_graph": 0}))
res.append(case("random 400 mixed", datasets.random_pose_graph(400, 350, 2), 3, {"use_graph": 0}))
res.append(case("random 400 BIG path only", datasets.random_pose_graph(400, 350, 2), 3, {"small_lds_kb": 0, "use_graph": 0}))
res.append(case("random 400 MEDIUM path only", datasets.random_pose_graph(400, 350, 2), 3, {"small_lds_kb": 0, "use_graph": 0}))
res.append(case("random 400 hipGraph", datasets.random_pose_graph(400, 350, 2), 3))
res.append(case("lattice 24", lib.lattice_arrays(24), 2, {"use_graph": 0}))
res.append(case("M3500 no graph", datasets.m3500_batch(), 3, {"use_graph": 0}))
res.append(case("M3500 hipGraph", datasets.m3500_batch(), 5))
res.append(case("M3500 BIG path only", datasets.m3500_batch(), 2, {"small_lds_kb": 0}))
res.append(case("M3500 MEDIUM path only", datasets.m3500_batch(), 2, {"small_lds_kb": 0}))
if not quick:
    res.append(case("lattice 60", lib.lattice_arrays(60), 2))
    res.append(case("random 1500", datasets.random_pose_graph(1500, 900, 3), 3))
print("SUMMARY:", sum(res), "/", len(res), "ok", flush=True)
sys.exit(0 if all(res) else 1)
