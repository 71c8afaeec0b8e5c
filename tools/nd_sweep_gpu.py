#!/usr/bin/env python3
"""measured (not modelled) effect of the nested-dissection knobs on the M3500 iteration time: python tools/nd_sweep_gpu.py"""
import os, subprocess, sys
code = r'''
import sys, os, time, ctypes as C
sys.path.insert(0, os.environ["ROOT"])
from aprilsam_amd import datasets, host
lib = host.SolverLib()
lib.set_option("leaf_nodes", int(os.environ["LEAF"]))
arr = datasets.m3500_batch()
g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
assert lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr) == 0
lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 20, 0); lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 200, 0); lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr)
    best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
st = p.stats()
print(f"{best:.4f} levels {st['n_levels']} fronts {st['n_fronts']} nnzL {st['nnz_L']} flops {st['flops_factor']:.3g}")
'''
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
grid = [(0.62, 25, 100, 8, 4, 2, 16)]
for leaf in (10, 12, 20, 24, 32):
    grid.append((0.62, 25, 100, 8, 4, 2, leaf))
for imb, lin, quad in ((0.58, 25, 100), (0.66, 25, 100), (0.62, 10, 100), (0.62, 40, 100), (0.62, 25, 50), (0.62, 25, 200), (0.60, 40, 200), (0.66, 10, 50)):
    grid.append((imb, lin, quad, 8, 4, 2, 16))
grid += [(0.62, 25, 100, 4, 4, 2, 16), (0.62, 25, 100, 16, 4, 2, 16), (0.62, 25, 100, 8, 2, 2, 16), (0.62, 25, 100, 8, 8, 2, 16), (0.62, 25, 100, 8, 4, 3, 16)]
for a in grid:
    imb, lin, quad, dirs, ref, band, leaf = a
    env = dict(os.environ, ROOT=ROOT, APRILSAM_AMD_ND_IMB=str(imb), APRILSAM_AMD_ND_LIN=str(lin), APRILSAM_AMD_ND_QUAD=str(quad), APRILSAM_AMD_ND_DIRS=str(dirs),
               APRILSAM_AMD_ND_REF=str(ref), APRILSAM_AMD_ND_BAND=str(band), LEAF=str(leaf))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(a, r.stdout.strip() or r.stderr[-300:], flush=True)
