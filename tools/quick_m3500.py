#!/usr/bin/env python3
"""M3500 batch: resident ms per iteration and warm API call, option on / off.   python tools/quick_m3500.py name [values...]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, host
lib = host.SolverLib()
arr = datasets.m3500_batch()
name = sys.argv[1] if len(sys.argv) > 1 else "lin_in_front"
vals = [float(v) for v in sys.argv[2:]] or [1, 0, 1, 0]
for v in vals:
    with lib.options(**{name: v}):
        g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
        assert lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr) == 0
        lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 20, 0); lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 200, 0); rc = lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr); best = min(best, (time.perf_counter() - t0) / 200)
        lib.dll.aprilsam_amd_resident_end(g.ptr, p.ptr); p.destroy(); g.destroy()
        g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
        for _ in range(10): g.cholesky(p)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); g.cholesky(p); ts.append(time.perf_counter() - t0)
        print(f"{name}={v:g}: resident {1e3 * best:.4f} ms/iter (rc {rc}), API call median {1e3 * np.median(ts):.4f} ms, chi2 {g.chi2():.6f}", flush=True)
        p.destroy(); g.destroy()
