"""Python counterparts of the reference's two example drivers, over the C-ABI (host.SolverLib):

* `run_tutorial`  — the 6-pose dog-leg scenario of examples/aprilsam_tutorial.c:80-266;
* `run_demo`      — the pose-by-pose simulator of examples/aprilsam_demo.c:119-234: per pose k append the
  node (state = init), add the prior at pose 0, copy every loaded factor whose max node id == k in file
  order, seed the new pose from "odom" factors (|a-b| == 1) with nb.state = na.state (+) z and
  relinearise it, then run the batch step (pose 0 or --batch_update_only) or the incremental step.

Both work with any library exporting the reference API (the product or oracle/_ref), which is how the
parity tests drive the two sides identically.  Timing = wall clock around the solver call only, the
region the demo times (aprilsam_demo.c:103-107); chi^2 is evaluated outside it.
"""
import math
import time

import numpy as np

from . import datasets


def _xyt_mul(a, b):      # doubles_xyt_mul, common/doubles_floats_impl.h:498-506
    s, c = math.sin(a[2]), math.cos(a[2])
    return [c * b[0] - s * b[1] + a[0], s * b[0] + c * b[1] + a[1], a[2] + b[2]]


def _xyt_inv(a):         # doubles_xyt_inv, common/doubles_floats_impl.h:569-575
    s, c = math.sin(a[2]), math.cos(a[2])
    return [-s * a[1] - c * a[0], -c * a[1] + s * a[0], -a[2]]


def _xyt_inv_mul(a, b):  # doubles_xyt_inv_mul, common/doubles_floats_impl.h:619-630
    c, s = math.cos(a[2]), math.sin(a[2])
    dx, dy = b[0] - a[0], b[1] - a[1]
    return [c * dx + s * dy, -s * dx + c * dy, b[2] - a[2]]


def run_tutorial(lib, batch_update_only=False, nthreshold=100, delta_xy=0.1, delta_theta=0.1, deterministic=True):
    """Returns list of (chi2, states[N,3]) after each of the 6 steps."""
    g = lib.new_graph(); p = lib.new_param(nthreshold=nthreshold, delta_xy=delta_xy, delta_theta=delta_theta)
    out = []
    Wodo = np.diag([1.0 / 0.1 ** 2, 1.0 / 0.1 ** 2, 1.0 / math.radians(1) ** 2]).reshape(9)

    def optimise(first):
        if first or batch_update_only:
            g.cholesky(p)
        else:
            if deterministic:
                p.c.batch_time = 1e300       # neutralises the wall-clock rule aprilsam.c:557 without patching
            g.cholesky_inc(p)
        out.append((g.chi2(), g.states()))

    g.add_node_xyt([0, 0, 0])
    g.add_factor_xytpos(0, [0, 0, 0], datasets.PRIOR_W)
    optimise(True)
    for k in range(1, 6):
        g.add_node_xyt([k, 0, 0])
        zz = _xyt_inv_mul([k - 1, 0, 0], [k, 0, 0])
        g.add_factor_xyt(k - 1, k, zz, Wodo)
        if k == 5:
            g.add_factor_xyt(0, 5, _xyt_inv_mul([0, 0, 0], [5, 1, 0]), Wodo)
        optimise(False)
    p.destroy(); g.destroy()
    return out


def run_demo(lib, arrays, batch_update_only=False, nthreshold=100, delta_xy=0.1, delta_theta=0.1,
             max_poses=None, deterministic=True, record_states_every=0, on_step=None, batch_every=0):
    """arrays = (states, fa, fb, z, W) of the LOADED graph (no prior). Returns dict with per-step chi2, ms,
    batch fall-back flags and the final states.  Beyond the demo: factors with fb < 0 are xytpos priors added with their pose;
    batch_every = k > 0 calls the batch step at every k-th pose (a caller-made fall-back)."""
    states, fa, fb, z, W = arrays
    N = len(states) if max_poses is None else min(max_poses, len(states))
    # factors grouped by the pose at which the demo adds them (max node id), file order preserved
    by_pose = [[] for _ in range(N)]
    for k in range(len(fa)):
        m = max(int(fa[k]), int(fb[k]))          # (a prior, fb = -1: its own pose)
        if m < N:
            by_pose[m].append(k)
    g = lib.new_graph(); p = lib.new_param(nthreshold=nthreshold, delta_xy=delta_xy, delta_theta=delta_theta)
    chi2 = np.zeros(N); ms = np.zeros(N); was_batch = np.zeros(N, bool)
    snaps = {}
    for k in range(N):
        g.add_node_xyt(states[k])
        if k == 0:
            g.add_factor_xytpos(0, [0, 0, 0], datasets.PRIOR_W)
        else:
            for f in by_pose[k]:
                a, b = int(fa[f]), int(fb[f])
                if b < 0:
                    g.add_factor_xytpos(a, z[f], W[f])
                    continue
                if abs(a - b) == 1:                      # "odom" (aprilsam_demo.c:83-87,172-191)
                    if a < b:
                        g.set_state(b, _xyt_mul(g.states_of(a), z[f]), relinearize=True)
                    else:
                        g.set_state(a, _xyt_mul(g.states_of(b), _xyt_inv(z[f])), relinearize=True)
                g.add_factor_xyt(a, b, z[f], W[f])
        n_before = p.c.factor_num
        t0 = time.perf_counter()
        if batch_update_only or k == 0 or (batch_every and k % batch_every == 0):
            g.cholesky(p); was_batch[k] = True
        else:
            if deterministic:
                p.c.batch_time = 1e300
            bt = p.c.batch_time
            g.cholesky_inc(p)
            was_batch[k] = p.c.batch_time != bt         # a fall-back batch rewrites batch_time
        ms[k] = (time.perf_counter() - t0) * 1e3
        if on_step is not None:
            on_step(k, p, bool(was_batch[k]))
        chi2[k] = g.chi2()
        if record_states_every and (k % record_states_every == 0 or k == N - 1):
            snaps[k] = g.states()
        del n_before
    final = g.states()
    p.destroy(); g.destroy()
    return dict(chi2=chi2, ms=ms, was_batch=was_batch, final_states=final, snaps=snaps)
