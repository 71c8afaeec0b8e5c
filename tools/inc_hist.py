import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from aprilsam_amd import datasets, harness, host
lib = host.SolverLib()
res = harness.run_demo(lib, datasets.m3500_arrays(), deterministic=True)
ms = res["ms"]; wb = res["was_batch"]
inc = ms[~wb]
print("inc steps", len(inc), "percentiles 1/10/25/50/75/90/99:", np.percentile(inc, [1, 10, 25, 50, 75, 90, 99]).round(3))
print("batch fallbacks", wb.sum(), "mean ms", ms[wb].mean().round(3), "total batch ms", ms[wb].sum().round(1), "total inc ms", inc.sum().round(1))
