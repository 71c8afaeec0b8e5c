#!/usr/bin/env python3
"""debug: phase stamps of k_block_chain for the root front of a lattice (APRILSAM_AMD_KPROF=3):
per panel p = 0..2, all taken by the CHAIN wave: chain start, chain end, past barrier 1 (inverse in LDS) and its own write-backs issued, past barrier 2
(the six row blocks solved against the panel), past barrier 3 (next diagonal block updated and handed over)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["APRILSAM_AMD_KPROF"] = "3"
from aprilsam_amd import host
K = int(sys.argv[1]) if len(sys.argv) > 1 else 316
lib = host.SolverLib()
g = lib.new_graph(); g.build_from_arrays(*lib.lattice_arrays(K)); p = lib.new_param()
lib.set_option("use_graph", 0)
for _ in range(2):
    g.cholesky(p)
nF = p.stats()["n_fronts"]
buf = np.zeros((nF, 16), np.int64)
lib.dll.aprilsam_amd_debug_front_times(p.ptr, buf.ctypes.data_as(C.POINTER(C.c_longlong)), nF)
for t in (nF - 1, nF - 2):
    b = buf[t]
    if b[0] == 0:
        continue
    print(f"front {t}: (us from kernel start, last outer block it ran)")
    for q in range(3):
        v = [(b[k + 5 * q] - b[0]) * 0.01 for k in range(1, 6)]
        print(f"  panel {q}: chain start {v[0]:6.2f}  chain end {v[1]:6.2f} (+{v[1]-v[0]:.2f})  barrier 1 + write-backs issued {v[2]:6.2f} (+{v[2]-v[1]:.2f})  barrier 2, rows solved {v[3]:6.2f} (+{v[3]-v[2]:.2f})  barrier 3, next block ready {v[4]:6.2f} (+{v[4]-v[3]:.2f})")
