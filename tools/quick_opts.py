#!/usr/bin/env python3
"""M3500 batch under option sets: resident ms per iteration, warm API call, cold call, chi^2 after 10 iterations vs the reference golden.
    python tools/quick_opts.py "amalg=1,amalg_max=48" "amalg=0" ...          (an empty spec = the defaults)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, host
lib = host.SolverLib()
arr = datasets.m3500_batch()
G = np.load(os.path.join(ROOT, "tests", "golden", "m3500_batch.npz"))
for spec in sys.argv[1:] or [""]:
    o = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in spec.split(",") if kv}
    with lib.options(**o):
        g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
        chi2, _ = g.batch_resident(p, 10)
        err = float(np.max(np.abs(chi2 - G["chi2"]) / G["chi2"])); serr = float(np.max(np.abs(g.states() - G["final_states"])))
        p.destroy(); g.destroy()
        g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
        assert lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr) == 0
        lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 20, 0); lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 200, 0); rc = lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr); best = min(best, (time.perf_counter() - t0) / 200)
        st = p.stats()
        lib.dll.aprilsam_amd_resident_end(g.ptr, p.ptr); p.destroy(); g.destroy()
        cold = []
        for _ in range(3):
            g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
            t0 = time.perf_counter(); g.cholesky(p); cold.append(time.perf_counter() - t0)
            if _ < 2: p.destroy(); g.destroy()
        for _ in range(10): g.cholesky(p)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); g.cholesky(p); ts.append(time.perf_counter() - t0)
        print(f"{spec or 'defaults'}: resident {1e3 * best:.4f} ms/iter (rc {rc}), API call median {1e3 * np.median(ts):.4f} ms, cold {1e3 * np.median(cold):.3f} ms, fronts {st['n_fronts']} levels {st['n_levels']} "
              f"Mflop {st['flops_factor'] / 1e6:.1f}, chi2 relerr vs golden {err:.2e} states {serr:.2e}", flush=True)
        p.destroy(); g.destroy()
