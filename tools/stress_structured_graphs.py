#!/usr/bin/env python3
"""structured sweep (minutes of GPU time; a bounded slice runs in tests/test_gpu_sweeps.py): chains, stars, complete graphs, two components, banded and
comb graphs against the oracle (default options / every front on the multi-workgroup path / tiny LDS budget and leaves).  python tools/stress_structured_graphs.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
from tests.support.oracle_binding import Oracle
from tests.support import sweeps
lib = host.SolverLib(); oracle = Oracle()
cases = [("chain", 5000), ("chain", 97), ("star", 1500), ("star", 130), ("complete", 60), ("complete", 140), ("two", 2000), ("band", 3000), ("band", 257), ("comb", 2500), ("comb", 4100)]
worst = sweeps.sweep_batch(lib, oracle, [(f"{k} n={n}", sweeps.structured(k, n, 300 + i)) for i, (k, n) in enumerate(cases)],
                           (dict(), dict(small_lds_kb=0), dict(small_lds_kb=16, leaf_nodes=6)), 1e-6, 1e-5, log=lambda s: print(s, flush=True))
print("all ok, worst chi2 relerr", worst)
