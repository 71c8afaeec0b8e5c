"""Scenario for SURVEY.md §8 row f2: a pose graph that mixes the two native factor types with two factor types neither
library knows (tests/support/custom_factor.c), solved through the reference API — 6 batch iterations, then two
incremental steps.  Run against the unmodified reference it produces tests/golden/custom_factors.npz
(oracle/gen_golden.py --custom); run against the product library it is the parity test."""
import ctypes as C
import os
import subprocess

import numpy as np

from aprilsam_amd import abi, datasets

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_custom_lib(outdir):
    out = os.path.join(outdir, "libcustom_factor.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "support", "custom_factor.c"), "-o", out, "-lm"])
    cl = C.CDLL(out)
    cl.custom_xy_create.restype = C.POINTER(abi.Factor)
    cl.custom_xy_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    cl.custom_heading_create.restype = C.POINTER(abi.Factor)
    cl.custom_heading_create.argtypes = [C.c_int, C.c_double, C.c_double]
    cl.custom_midpoint_create.restype = C.POINTER(abi.Factor)
    cl.custom_midpoint_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return cl


def _rel_xy(pa, pb):
    c, s = np.cos(pa[2]), np.sin(pa[2])
    dx, dy = pb[0] - pa[0], pb[1] - pa[1]
    return np.array([c * dx + s * dy, -s * dx + c * dy])


def run(lib, cl, batch_iters=6, triples=0):
    """lib: host.SolverLib (reference or product); cl: the compiled custom-factor helper.  Returns a dict of arrays.
    triples > 0: that many three-pose "midpoint" factors (type 79, factor->nnodes == 3) join the batch graph and one more each
    of the two incremental steps (golden: custom_factors3.npz)."""
    rng = np.random.default_rng(1234)
    states, fa, fb, z, W = datasets.random_pose_graph(40, 25, seed=5)
    truth = states.copy()
    g = lib.new_graph(); g.build_from_arrays(states, fa, fb, z, W); p = lib.new_param(nthreshold=1000)

    def add_xy(a, b):
        zz = _rel_xy(truth_all[a], truth_all[b]) + rng.normal(0, 0.05, 2)
        M = rng.normal(size=(2, 2)); Wm = M @ M.T + np.diag([30.0, 30.0])
        f = cl.custom_xy_create(int(a), int(b), (C.c_double * 2)(*zz), (C.c_double * 4)(*Wm.reshape(4)))
        lib._add_factor(g.ptr, f)

    def add_heading(a):
        f = cl.custom_heading_create(int(a), float(truth_all[a][2] + rng.normal(0, 0.02)), float(rng.uniform(50, 200)))
        lib._add_factor(g.ptr, f)

    truth_all = [t for t in truth]
    pairs = []
    while len(pairs) < 12:
        a, b = rng.integers(0, 40, 2)
        if a != b and (a, b) not in pairs:
            pairs.append((int(a), int(b)))
    for a, b in pairs:
        add_xy(a, b)
    for a in rng.choice(40, 6, replace=False):
        add_heading(int(a))

    def add_mid(a, b, c):
        pa, pb, pc = truth_all[a], truth_all[b], truth_all[c]
        d = pb[:2] - 0.5 * (pa[:2] + pc[:2]); cc, ss = np.cos(pa[2]), np.sin(pa[2])
        zz = np.array([cc * d[0] + ss * d[1], -ss * d[0] + cc * d[1]]) + rng3.normal(0, 0.05, 2)
        M = rng3.normal(size=(2, 2)); Wm = M @ M.T + np.diag([20.0, 20.0])
        lib._add_factor(g.ptr, cl.custom_midpoint_create(int(a), int(b), int(c), (C.c_double * 2)(*zz), (C.c_double * 4)(*Wm.reshape(4))))

    rng3 = np.random.default_rng(4321)                    # (its own stream: the scenario without triples stays what the first golden pins)
    for _ in range(triples):
        a, b, c = (int(v) for v in rng3.choice(40, 3, replace=False))
        add_mid(a, b, c)
    # perturb the start so that Gauss-Newton has work to do
    for i in range(1, 40):
        g.set_state(i, truth[i] + np.array([rng.normal(0, 0.3), rng.normal(0, 0.3), rng.normal(0, 0.08)]))
    chi2 = [g.chi2()]
    batch_states = []
    for _ in range(batch_iters):
        g.cholesky(p)
        chi2.append(g.chi2()); batch_states.append(g.states())
    # two incremental steps: a new pose with an odometry factor + both foreign types, then a native-only step
    inc_states = []
    for step in range(2):
        last = g.n_nodes - 1
        pl = np.array(g.states_of(last))
        new_truth = np.array([pl[0] + np.cos(pl[2]), pl[1] + np.sin(pl[2]), pl[2] + 0.1])
        truth_all.append(new_truth)
        n = g.add_node_xyt(new_truth + np.array([0.05, -0.04, 0.01]))
        g.add_factor_xyt(last, n, [1.0, 0.0, 0.1], np.diag([100.0, 100.0, 400.0]))
        if step == 0:
            add_xy(10, n); add_heading(n)
        if triples:
            add_mid(n, int(rng3.integers(0, 20)), int(rng3.integers(20, 40))) if step == 0 else add_mid(int(rng3.integers(0, 40)), n - 1, n)
        g.cholesky_inc(p)
        chi2.append(g.chi2()); inc_states.append(g.states())
    out = dict(chi2=np.array(chi2), batch_states=np.array(batch_states), inc_states_0=inc_states[0], inc_states_1=inc_states[1],
               n_factors=np.array(g.n_factors))
    p.destroy(); g.destroy()
    return out
