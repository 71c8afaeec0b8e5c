#!/usr/bin/env python3
"""Short bench summary for optimisation loops: python tools/quick_bench.py [--lattice]"""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--no-cpu-baseline", "--no-inc"]
if "--lattice" not in sys.argv:
    args.append("--no-lattice")
r = subprocess.run(args, capture_output=True, text=True)
try:
    d = json.loads(r.stdout.strip().splitlines()[-1])
except Exception:
    print(r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
print("M3500 ms_per_step %.4f  it/s %.1f  factorise_ms %.4f  parity %.2e  first_call %.2f ms" % (
    d["ms_per_step"], d["value"], d["factorise_ms"], d["parity"]["chi2_max_relerr_10_iters"], d["first_call_ms_incl_symbolic"]))
print("  kernels ms/step:", d["kernels_ms_per_step"])
print("  launches/step  :", d["kernel_launches_per_step"])
print("  roofline:", {k: d["roofline"][k] for k in ("kernel", "achieved", "unit", "frac", "avg_launch_us")})
if "api" in d:
    print("  api:", json.dumps(d["api"]))
if "lattice100k" in d:
    L = d["lattice100k"]
    print("lattice100k:", {k: L[k] for k in L if k != "workload"})
