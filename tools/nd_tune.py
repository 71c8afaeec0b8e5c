"""Random search over the nested-dissection scoring knobs (APRILSAM_AMD_ND_*) against a per-level cost model of the M3500\niteration (us per level = fixed + pivots + children + area, + back substitution); host only.  python tools/nd_tune.py"""
import os, sys, subprocess, random
from concurrent.futures import ThreadPoolExecutor
code = r'''
import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from aprilsam_amd import datasets, host
from tests.support.mf_emulator import PlanView
lib = host.SolverLib()
s, fa, fb, z, W = datasets.m3500_batch()
P = PlanView(lib, len(s), fa, fb, xy=s[:, :2].copy(), leaf_nodes=int(os.environ["LEAF"]))
nub = P.front_nub; nsb = P.front_nsb; lev = P.front_level
nch = np.diff(P.ch_ptr)
cost = 0.0
for l in range(lev.max() + 1):
    idx = np.where(lev == l)[0]
    R = 3 * (nsb[idx] + nub[idx] + 1)
    t = 8 + 1.25 * nsb[idx] + 2.0 * nch[idx] + 0.00035 * R * R
    cost += t.max() + 10
print(lev.max() + 1, int(P.stats[3]), round(cost, 1))
'''
random.seed(5)
def run(a):
    imb, lin, quad, dirs, ref, band, leaf = a
    env = dict(os.environ, APRILSAM_AMD_ND_IMB=str(imb), APRILSAM_AMD_ND_LIN=str(lin), APRILSAM_AMD_ND_QUAD=str(quad), APRILSAM_AMD_ND_DIRS=str(dirs),
               APRILSAM_AMD_ND_REF=str(ref), APRILSAM_AMD_ND_BAND=str(band), LEAF=str(leaf), APRILSAM_AMD_PLAN_THREADS="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.split()
    return (float(out[2]),) + a + (int(out[0]), int(out[1]))
grid = [(round(random.uniform(0.56, 0.70), 3), random.choice([0, 5, 10, 25, 40]), random.choice([0, 20, 50, 100, 150, 250]), random.choice([2, 4, 8, 16]),
         random.choice([2, 4, 8]), random.choice([1, 2, 3]), random.choice([12, 14, 16, 18, 20])) for _ in range(140)]
grid.append((0.62, 25, 100, 8, 4, 2, 16))
with ThreadPoolExecutor(6) as ex: res = list(ex.map(run, grid))
for r in sorted(res)[:12]: print(r)
