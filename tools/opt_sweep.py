#!/usr/bin/env python3
"""option sweeps on a lattice: ms per iteration (resident loop), kernel times, sum c_j^2.   python tools/opt_sweep.py K "a=1,b=2" "a=3" ..."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
lib = host.SolverLib()
lib.dll.aprilsam_amd_resident_chi2.restype = C.c_double
K = int(sys.argv[1])
for spec in sys.argv[2:] or [""]:
    o = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in spec.split(",") if kv}
    with lib.options(**o):
        g = lib.new_graph(); lib.dll.aprilsam_amd_make_lattice(g.ptr, K); p = lib.new_param()
        t0 = time.time()
        assert lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr) == 0
        setup = time.time() - t0
        lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 1, 0); assert lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr) == 0
        chi = lib.dll.aprilsam_amd_resident_chi2(g.ptr)
        n = 10 if K < 500 else 3
        t0 = time.time(); lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, n, 0); rc = lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr); dt = (time.time() - t0) / n
        lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 2, 1)
        ms = (C.c_double * 16)(); calls = (C.c_longlong * 16)(); fl = (C.c_double * 16)(); by = (C.c_double * 16)(); names = (C.c_char_p * 16)()
        nk = lib.dll.aprilsam_amd_kernel_profile(p.ptr, ms, calls, fl, by, names)
        st = p.stats()
        prof = {names[k].decode()[2:]: round(ms[k] / 2, 3) for k in range(nk) if ms[k] > 0}
        print(f"K={K} {spec or 'defaults'}: {1e3 * dt:.3f} ms/iter rc {rc} chi2 {chi:.6e} sum_cj2 {st['flops_factor'] / 1e9:.1f} G fronts {st['n_fronts']} levels {st['n_levels']} setup {setup:.1f}s {prof}", flush=True)
        lib.dll.aprilsam_amd_resident_end(g.ptr, p.ptr); p.destroy(); g.destroy()
