import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from aprilsam_amd import datasets, harness, host
lib = host.SolverLib()
G = np.load("tests/golden/m3500_inc_demo.npz")
n = 3500
arr = datasets.m3500_arrays()
res = harness.run_demo(lib, arr, max_poses=n, deterministic=True)
rel = np.abs(res["chi2"] - G["chi2"][:n]) / np.maximum(G["chi2"][:n], 1e-9)
bad = np.nonzero(rel > 1e-8)[0]
print("first bad steps:", bad[:20].tolist(), "count", len(bad))
states, fa, fb, z, W = arr
for b in bad[:8]:
    fs = [(int(fa[k]), int(fb[k])) for k in range(len(fa)) if max(fa[k], fb[k]) == b]
    prev = [(int(fa[k]), int(fb[k])) for k in range(len(fa)) if max(fa[k], fb[k]) == b - 1]
    lastb = np.nonzero(res["was_batch"][:b + 1])[0][-1]
    print("step", b, "rel %.3e" % rel[b], "factors", fs, "prev step factors", prev, "last batch at", lastb)
