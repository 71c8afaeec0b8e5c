#!/usr/bin/env python3
"""randomised / structured sweep, not part of the test suite (minutes of GPU time): odd lattices sharded over 2 / 4 / 8 ranks on one GPU (host-callback transport) against the single-GPU states.  python tools/stress_sharded.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if __name__ == "__main__":
    from aprilsam_amd import host
    import tests.test_gpu_shard as S
    lib = host.SolverLib()
    for K in (37, 91, 131):
        c1, st1 = S._single_gpu_states(lib, K, 2)
        for world in (2, 4, 8):
            res, st = S._run(world, K, 2, timeout=300)
            chi2 = res[0][1]
            e = float(np.max(np.abs(st - st1))); ec = max(abs(a - b) / max(b, 1e-9) for a, b in zip(chi2, c1))
            print(f"K={K} world={world}: states vs single GPU {e:.2e} chi2 {ec:.2e} digests equal {len(set(r[6] for r in res)) == 1}", flush=True)
            assert e < 1e-9 and ec < 1e-9 and len(set(r[6] for r in res)) == 1
    print("all ok")
