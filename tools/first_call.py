#!/usr/bin/env python3
"""What the first solver call of a process pays: python tools/first_call.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
from aprilsam_amd import datasets, host
lib = host.SolverLib()
t1 = time.perf_counter(); print(f"import + dlopen {1e3*(t1-t0):.1f} ms")
if "--count-first" in sys.argv:
    n = lib.dll.aprilsam_amd_device_count(); t2 = time.perf_counter(); print(f"device_count() = {n}: {1e3*(t2-t1):.1f} ms")
arr = datasets.m3500_arrays()
s, fa, fb, z, W = arr
g = lib.new_graph(); p = lib.new_param()
g.add_node_xyt(list(s[0])); g.add_factor_xytpos(0, [0, 0, 0], datasets.PRIOR_W)
os.environ.get("X")
ta = time.perf_counter(); g.cholesky(p); tb = time.perf_counter(); print(f"first april_graph_cholesky (1 pose): {1e3*(tb-ta):.1f} ms")
ta = time.perf_counter(); g.cholesky(p); tb = time.perf_counter(); print(f"second: {1e3*(tb-ta):.3f} ms")
g2 = lib.new_graph(); g2.build_from_arrays(*datasets.m3500_batch()); p2 = lib.new_param()
ta = time.perf_counter(); g2.cholesky(p2); tb = time.perf_counter(); print(f"first call on M3500 (new param, warm process): {1e3*(tb-ta):.2f} ms")
ta = time.perf_counter(); g2.cholesky(p2); tb = time.perf_counter(); print(f"second: {1e3*(tb-ta):.3f} ms")
if "--chi2" in sys.argv:
    print(f"chi2 after two M3500 calls: {g2.chi2()!r}")
