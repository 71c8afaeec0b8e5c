#!/usr/bin/env python3
"""april_graph_chi2 on the device (k_chi2 + the deterministic sum): python tools/chi2_time.py [K ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
lib = host.SolverLib()
for K in [int(a) for a in sys.argv[1:]] or [60, 316, 1000]:
    g = lib.new_graph(); nfac = lib.dll.aprilsam_amd_make_lattice(g.ptr, K); p = lib.new_param()
    lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr)
    c0 = lib.dll.aprilsam_amd_resident_chi2(g.ptr)
    n = 20; t0 = time.perf_counter()
    for _ in range(n): c = lib.dll.aprilsam_amd_resident_chi2(g.ptr)
    dt = (time.perf_counter() - t0) / n
    assert c == c0
    print(f"K={K}: {nfac} factors, chi2 {c!r}: {dt*1e3:.3f} ms per call (kernels + 8-byte copy + sync)", flush=True)
    lib.dll.aprilsam_amd_resident_end(g.ptr, p.ptr); p.destroy(); g.destroy()
