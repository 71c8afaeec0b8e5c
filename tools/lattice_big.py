#!/usr/bin/env python3
"""single-GPU timing of a big lattice: python tools/lattice_big.py K [iters]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
K = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = host.SolverLib()
t0 = time.time(); g = lib.new_graph(); nfac = lib.dll.aprilsam_amd_make_lattice(g.ptr, K); p = lib.new_param()
print(f"K={K}: {g.n_nodes} poses, {nfac} factors, graph built in {time.time()-t0:.1f} s", flush=True)
t0 = time.time(); rc = lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr); print(f"resident_begin rc={rc} (pack + ordering + symbolic + upload) {time.time()-t0:.2f} s", flush=True)
chi = [lib.dll.aprilsam_amd_resident_chi2(g.ptr)]
for it in range(iters):
    t0 = time.time(); lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 1, 0); rc = lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr); dt = time.time() - t0
    chi.append(lib.dll.aprilsam_amd_resident_chi2(g.ptr))
    print(f"iter {it}: {dt*1e3:.1f} ms rc={rc} chi2 {chi[-1]:.6f}", flush=True)
st = p.stats(); print({k: st[k] for k in ("n_fronts", "n_levels", "max_front_rows", "nnz_L", "flops_factor", "bytes_fronts")})
lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 2, 1)
ms = (C.c_double * 16)(); calls = (C.c_longlong * 16)(); fl = (C.c_double * 16)(); by = (C.c_double * 16)(); names = (C.c_char_p * 16)()
n = lib.dll.aprilsam_amd_kernel_profile(p.ptr, ms, calls, fl, by, names)
print({names[k].decode(): round(ms[k] / 2, 3) for k in range(n)})
tf = sum(ms[k] for k in range(n) if names[k].decode() in ("k_front_small", "k_assemble_big", "k_panel_big", "k_syrk_big")) / 2
print(f"factor {tf:.1f} ms -> {st['flops_factor']/tf/1e9:.2f} TFLOP/s")
lv = (C.c_double * (6 * 64))()
lib.dll.aprilsam_amd_level_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
nl = lib.dll.aprilsam_amd_level_profile(C.cast(p.ptr, C.c_void_p), lv, 64)
print("level: fronts (multi-workgroup)  widest own part  GFLOP   factor ms   back-substitution ms      [per iteration]")
for l in range(nl):
    up, dn, nf, nb, mx, fl = (lv[6 * l + k] for k in range(6))
    print(f"  {l:2d}: {int(nf):7d} ({int(nb):3d})  {int(mx):6d}  {fl / 1e9:9.3f}  {up / 2:9.3f}  {dn / 2:9.3f}")
