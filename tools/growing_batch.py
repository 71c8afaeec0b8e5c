#!/usr/bin/env python3
"""the reference demo's --batch_update_only mode (one april_graph_cholesky per new pose on a growing graph): per-step time
of this library (batch_extend on / off) and of the reference on the same host.  python tools/growing_batch.py [n_poses]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host
from tests.support.oracle_binding import REFLIB
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
lib = host.SolverLib()
out = {"workload": f"M3500 demo --batch_update_only, first {n} poses: one april_graph_cholesky call per new pose (every call sees a new topology)"}
for ext in (1, 0):
    lib.set_option("batch_extend", ext)
    r = harness.run_demo(lib, datasets.m3500_arrays(), batch_update_only=True, max_poses=n)
    ms = r["ms"][1:]
    out["batch_extend_%d" % ext] = {"total_ms": float(ms.sum()), "mean_ms": float(ms.mean()), "median_ms": float(np.median(ms)), "p99_ms": float(np.percentile(ms, 99)),
                                    "final_chi2": float(r["chi2"][-1])}
lib.set_option("batch_extend", 1)
if os.path.exists(REFLIB) and "--no-ref" not in sys.argv:
    ref = host.SolverLib(REFLIB)
    r = harness.run_demo(ref, datasets.m3500_arrays(), batch_update_only=True, max_poses=n)
    ms = r["ms"][1:]
    out["reference_cpu_same_host"] = {"total_ms": float(ms.sum()), "mean_ms": float(ms.mean()), "median_ms": float(np.median(ms)), "p99_ms": float(np.percentile(ms, 99)),
                                      "final_chi2": float(r["chi2"][-1]), "cores": 1}
    out["speedup_total_vs_reference"] = out["reference_cpu_same_host"]["total_ms"] / out["batch_extend_1"]["total_ms"]
print(json.dumps(out, indent=1))
