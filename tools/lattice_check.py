#!/usr/bin/env python3
"""a lattice of any size on one GPU, checked against the normal equations (tests/support/normal_eq.py): python tools/lattice_check.py K [iters]
-- beyond config 5's K = 1000 this is a scale probe (index widths, memory): K = 2000 is 4 * 10^6 poses / 16 * 10^6 factors"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
from tests.support.normal_eq import normal_equation_residual
K = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = host.SolverLib()
t0 = time.time(); arr = lib.lattice_arrays(K); g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
print(f"K={K}: {g.n_nodes} poses, {len(arr[1])} factors, graph built in {time.time() - t0:.1f} s", flush=True)
t0 = time.time(); rc = lib.dll.aprilsam_amd_resident_begin(g.ptr, p.ptr)
print(f"resident_begin rc={rc} ({time.time() - t0:.2f} s: pack + ordering + symbolic + upload)", lib.last_error() if rc else "", flush=True)
if rc:
    sys.exit(1)
chi = [lib.dll.aprilsam_amd_resident_chi2(g.ptr)]
for it in range(iters):
    t0 = time.time(); lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 1, 0); rc = lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr); dt = time.time() - t0
    chi.append(lib.dll.aprilsam_amd_resident_chi2(g.ptr))
    print(f"iteration {it}: {1e3 * dt:.1f} ms rc={rc} chi2 {chi[-2]:.6e} -> {chi[-1]:.6e}", flush=True)
t0 = time.time(); lib.dll.aprilsam_amd_resident_steps(g.ptr, p.ptr, 2, 0); rc = lib.dll.aprilsam_amd_resident_sync(g.ptr, p.ptr)
print(f"two more, timed together: {1e3 * (time.time() - t0) / 2:.1f} ms per iteration rc={rc}", flush=True)
st = p.stats(); print({k: st[k] for k in ("n_fronts", "n_levels", "max_front_rows", "nnz_L", "flops_factor", "bytes_fronts")}, flush=True)
assert lib.dll.aprilsam_amd_resident_end(g.ptr, p.ptr) == 0
t0 = time.time()
out = normal_equation_residual(g.l_points(), arr[1], arr[2], arr[3], arr[4], g.deltas(), 1e-4)
print(f"normal equations of the last iteration: {out}  ({time.time() - t0:.1f} s of numpy)")
