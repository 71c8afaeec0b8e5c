"""debug: per-level phase breakdown of k_backsolve (sets APRILSAM_AMD_KPROF=2): python tools/backsolve_times.py [--lattice K]
(gather = x of the struct rows in LDS; first products = up to the first block's barrier; rest = the remaining blocks / the chain)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
os.environ["APRILSAM_AMD_KPROF"] = "2"
from aprilsam_amd import datasets, host
from tests.support.mf_emulator import PlanView
lib = host.SolverLib()
arr = lib.lattice_arrays(int(sys.argv[sys.argv.index("--lattice") + 1])) if "--lattice" in sys.argv else datasets.m3500_batch()
g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
lib.set_option("use_graph", 0)
for _ in range(3): g.cholesky(p)
nF = p.stats()["n_fronts"]
buf = np.zeros((nF, 16), np.int64)
lib.dll.aprilsam_amd_debug_front_times(p.ptr, buf.ctypes.data_as(C.POINTER(C.c_longlong)), nF)
P = PlanView(lib, len(arr[0]), arr[1], arr[2], xy=arr[0][:, :2], leaf_nodes=16)
for l in range(P.nLevels - 1, -1, -1):
    fr = P.lev_fronts[P.lev_ptr[l]:P.lev_ptr[l + 1]]
    big = max(fr, key=lambda t: buf[t, 7] - buf[t, 4])
    b = buf[big] * 0.01
    span = (max(buf[t, 7] for t in fr) - min(buf[t, 4] for t in fr)) * 0.01
    print(f"level {l}: span {span:.1f} | slowest nsb={P.front_nsb[big]} nub={P.front_nub[big]}: gather {b[5]-b[4]:.2f} first products {b[6]-b[5]:.2f} rest {b[7]-b[6]:.2f} total {b[7]-b[4]:.2f}")
