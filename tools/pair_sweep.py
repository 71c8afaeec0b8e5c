#!/usr/bin/env python3
"""options syrk_pair_tiles / syrk_group (big fronts: one wide update per GROUP of outer blocks, K = 256 / 384 / 512) on the lattices: ms per iteration, kernel times, chi^2.
python tools/pair_sweep.py [K ...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
lib = host.SolverLib()
lib.dll.aprilsam_amd_resident_chi2.restype = C.c_double
Ks = [int(a) for a in sys.argv[1:]] or [316, 1000]
for K in Ks:
    for pt, grp in ((0, 2), (2048, 2), (2048, 3), (2048, 4), (2048, 6), (512, 4)):
        with lib.options(syrk_pair_tiles=pt, syrk_group=grp):
            g = lib.new_graph(); lib.dll.aprilsam_amd_make_lattice(g.ptr, K); p = lib.new_param()
            assert lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr) == 0
            lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 1, 0); assert lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr) == 0
            chi = lib.dll.aprilsam_amd_resident_chi2(g.ptr)
            n = 10 if K < 500 else 3
            t0 = time.time(); lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, n, 0); rc = lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr); dt = (time.time() - t0) / n
            lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 2, 1)
            ms = (C.c_double * 16)(); calls = (C.c_longlong * 16)(); fl = (C.c_double * 16)(); by = (C.c_double * 16)(); names = (C.c_char_p * 16)()
            nk = lib.dll.aprilsam_amd_kernel_profile(p.ptr, ms, calls, fl, by, names)
            prof = {names[k].decode(): (round(ms[k] / 2, 3), int(calls[k] // 2)) for k in range(nk) if ms[k] > 0}
            sy = prof.get("k_syrk_big", (0, 0))[0]
            tf = fl[[names[k].decode() for k in range(nk)].index("k_syrk_big")] / (sy * 1e-3) / 1e12 if sy else 0
            print(f"K={K} syrk_pair_tiles={pt} syrk_group={grp}: {1e3 * dt:.3f} ms/iter rc {rc} chi2 after 1 {chi:.9e}  k_syrk_big {sy} ms ({tf:.1f} TFLOP/s)  {prof}", flush=True)
            lib.dll.aprilsam_amd_resident_end(g.ptr, p.ptr); p.destroy(); g.destroy()
