#!/usr/bin/env python3
"""structured sweep (minutes of GPU time; a bounded slice runs in tests/test_gpu_sweeps.py): lattices of odd sizes (K = 37 .. 170) against the oracle, four option sets.  python tools/stress_lattices.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
from tests.support.oracle_binding import Oracle
from tests.support import sweeps
lib = host.SolverLib(); oracle = Oracle()
sweeps.sweep_batch(lib, oracle, [(f"lattice K={K}", lib.lattice_arrays(K)) for K in (37, 91, 131, 170)],
                   (dict(), dict(small_lds_kb=0), dict(small_lds_kb=48, leaf_nodes=24), dict(small_lds_kb=0, leaf_nodes=7, syrk_xcd_order=1, syrk_small_tiles=1 << 30)),
                   1e-8, 1e-6, log=lambda s: print(s, flush=True))
print("all ok")
