#!/bin/bash
# how many regenerated fronts / walked fronts a single-workgroup step (k_inc_one) should take on: M3500 demo totals
IFS=";" read -ra LIST <<< "${CFGS:-3 4;5 6;8 10;2 3;3 4;5 6}"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  echo -n "inc_one_up $1 inc_one_dn $2: "
  APRILSAM_AMD_INC_ONE_UP=$1 APRILSAM_AMD_INC_ONE_DN=$2 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from aprilsam_amd import datasets, harness, host
lib = host.SolverLib()
arr = datasets.m3500_arrays()
harness.run_demo(lib, arr, max_poses=300, deterministic=True)
best = None
for _ in range(3):
    res = harness.run_demo(lib, arr, deterministic=True)
    ms = res["ms"]
    t = (ms.sum(), np.median(ms))
    best = t if best is None or t[0] < best[0] else best
print(f"total {best[0]:.1f} ms median {best[1]:.4f}")
PY
done
