#!/usr/bin/env python3
"""randomised sweep of the batch path against the oracle: pose graphs of 300-3 600 poses with 0.3-1.6 random loop closures per pose (dense root
fronts of every width modulo the 32- and 128-column blockings), default options / every front on the multi-workgroup path / without the wide
back substitution.  Found the out-of-bounds staging reads of round 3's k_block_solve and k_backsolve_blk.  A bounded slice runs in
tests/test_gpu_sweeps.py.   python tools/stress_random_graphs.py [seed base] [pool_guard doubles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import host
from tests.support.oracle_binding import Oracle
from tests.support import sweeps
lib = host.SolverLib(); oracle = Oracle()
base = int(sys.argv[1]) if len(sys.argv) > 1 else 7          # another argument: another set of graphs
guard = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with lib.options(pool_guard=guard):
    worst = sweeps.sweep_batch(lib, oracle, sweeps.random_graph_cases(base, 24), (dict(), dict(small_lds_kb=0), dict(small_lds_kb=0, blk_backsolve=0)), 1e-7, 1e-5,
                               log=lambda s: print(s, flush=True))
print("all ok, worst chi2 relerr", worst)
