#!/usr/bin/env python3
"""Config 3 driver: the M3500 pose-by-pose demo (examples/aprilsam_demo.c semantics) on the GPU library,
compared with the reference golden.  python tools/inc_demo.py [n_poses] [--ref]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3500
lib = host.SolverLib()
G = np.load(os.path.join(ROOT, "tests", "golden", "m3500_inc_demo.npz"))
t0 = time.time()
res = harness.run_demo(lib, datasets.m3500_arrays(), max_poses=n, deterministic=True)
wall = time.time() - t0
rel = np.abs(res["chi2"] - G["chi2"][:n]) / np.maximum(G["chi2"][:n], 1e-9)
ms = res["ms"]
print(f"poses {n}: solver total {ms.sum():.1f} ms  mean {ms.mean():.3f}  median {np.median(ms):.3f}  p99 {np.percentile(ms, 99):.3f}  max {ms.max():.2f} ms   (wall incl. python {wall:.1f} s)")
rms = G["ref_ms"][:n]
print(f"reference CPU (survey container): total {rms.sum():.1f} ms mean {rms.mean():.3f} median {np.median(rms):.3f} p99 {np.percentile(rms, 99):.3f}")
print("fallbacks equal:", np.array_equal(res["was_batch"], G["was_batch"][:n]), " ours at", (np.nonzero(res["was_batch"])[0] + 1)[:12].tolist())
print(f"chi2 max rel err {rel.max():.3e} at step {int(rel.argmax())}; final chi2 {res['chi2'][-1]:.9f} (ref {G['chi2'][n-1]:.9f})")
if n == 3500:
    print("max |state - ref|:", np.max(np.abs(res["final_states"] - G["final_states"])))
