#!/usr/bin/env python3
"""debug: per-level phase breakdown of k_front_small (needs APRILSAM_AMD_KPROF=1)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["APRILSAM_AMD_KPROF"] = "1"
from aprilsam_amd import datasets, host
from tests.support.mf_emulator import PlanView
lib = host.SolverLib()
arr = datasets.m3500_batch() if "--lattice" not in sys.argv else lib.lattice_arrays(316)
g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
lib.set_option("use_graph", 0)
for _ in range(3):
    g.cholesky(p)
st = p.stats(); nF = st["n_fronts"]
buf = np.zeros((nF, 16), np.int64)
n = lib.dll.aprilsam_amd_debug_front_times(p.ptr, buf.ctypes.data_as(C.POINTER(C.c_longlong)), nF)
print("fronts", n)
perm = np.array([p.c.ordering[i] for i in range(len(arr[0]))])
P = PlanView(lib, len(arr[0]), arr[1], arr[2], xy=arr[0][:, :2], leaf_nodes=16)
T = buf[:, :4] * 0.01   # us
for l in range(P.nLevels):
    fr = P.lev_fronts[P.lev_ptr[l]:P.lev_ptr[l + 1]]
    fr = [t for t in fr if T[t, 0] > 0]
    if not fr:
        print(f"level {l}: no small fronts"); continue
    a = np.array([[T[t, 1] - T[t, 0], T[t, 2] - T[t, 1], T[t, 3] - T[t, 2]] for t in fr])
    span = max(T[t, 3] for t in fr) - min(T[t, 0] for t in fr)
    big = max(fr, key=lambda t: T[t, 3] - T[t, 0])
    print(f"level {l}: {len(fr):4d} fronts  span {span:7.1f} us | mean asm {a[:,0].mean():6.1f} fac {a[:,1].mean():6.1f} store {a[:,2].mean():6.1f} | "
          f"slowest nsb={P.front_nsb[big]} nub={P.front_nub[big]} nch={P.ch_ptr[big+1]-P.ch_ptr[big]}: asm {T[big,1]-T[big,0]:.1f} fac {T[big,2]-T[big,1]:.1f} store {T[big,3]-T[big,2]:.1f} | chain+far {buf[big,8]*0.01:.1f} next-block syrk {buf[big,9]*0.01:.1f} (pure look-ahead chains {buf[big,15]*0.01:.2f} us = {buf[big,10]} cycles) | asm: zero {(buf[big,4]-buf[big,0])*0.01:.1f} dest {(buf[big,5]-buf[big,4])*0.01:.1f} fill {(buf[big,6]-buf[big,5])*0.01:.1f} (n={buf[big,7]}) rest {(buf[big,1]-buf[big,6])*0.01:.1f} (after wait {(buf[big,1]-buf[big,11])*0.01 if buf[big,11] else 0:.1f})")
