#!/usr/bin/env python3
"""host-side planning time (ordering + symbolic, no GPU involved): python tools/plan_time.py  [set APRILSAM_AMD_PLAN_THREADS=1 to compare]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, host
from tests.support.mf_emulator import PlanView
lib = host.SolverLib()
cases = [("M3500", datasets.m3500_batch())] + [(f"lattice {k}x{k}", lib.lattice_arrays(k)) for k in (100, 316)]
for name, (s, fa, fb, z, W) in cases:
    xy = np.ascontiguousarray(s[:, :2])
    fn = np.ascontiguousarray(np.column_stack([fa, fb]).astype(np.int32))
    import ctypes as C
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        h = lib.dll.aprilsam_amd_plan_create(len(s), len(fa), fn.ctypes.data_as(C.POINTER(C.c_int)), xy.ctypes.data_as(C.POINTER(C.c_double)), 16)
        best = min(best, time.perf_counter() - t0)
        lib.dll.aprilsam_amd_plan_destroy(h)
    print(f"{name}: plan_create {best*1e3:.2f} ms  (threads env: {os.environ.get('APRILSAM_AMD_PLAN_THREADS', 'auto')})")
