"""Residual of the Gauss-Newton normal equations, matrix-free, in numpy.  TEST INFRASTRUCTURE (a checker, like oracle/):
used by tests/ and by bench.py's parity legs outside every timed region -- never by the product package.

What it checks is independent of ANY factorisation, ordering or recorded trace: for the linearisation point lp of a step
and the dx the solver returned,

    res = sum_f J_f^T W_f (J_f dx_f - r_f) + lambda dx   =   (J^T W J + lambda I) dx - J^T W r        (aprilsam.c:141-204, 233-262)

with J_f, r_f the Jacobians / residual of factor f at lp (april_graph_xyt.c:62-124 for xyt, april_graph_xytpos.c:63-102 for
the prior) -- the system the reference assembles and solves.  A correct dx leaves |res| at rounding level relative to
the terms of J^T W r (their sum without cancellation: near convergence J^T W r itself goes to zero); a wrong pivot, a lost
update block or a stale front leaves it at order one.  Cost O(F): usable at the full sizes
of BASELINE.json's configs 4 and 5 (10^5 and 10^6 poses), where the reference itself cannot run.

Valid for information matrices that are symmetric as given (with an asymmetric W the reference's system depends on its own
elimination order, tests/test_gpu_asymmetric_w.py)."""
import numpy as np

TWOPI = 6.2831853071795862319959


def mod2pi(v):                      # common/math_util.h:113-122, range [-pi, pi)
    vin = v + np.pi
    return vin - TWOPI * np.floor(vin / TWOPI) - np.pi


def linearise(lp, fa, fb, z):
    """per factor: J_a [F,3,3], J_b [F,3,3] (zero for priors), r [F,3] at the linearisation point lp [N,3]"""
    lp = np.asarray(lp, float); z = np.asarray(z, float).reshape(-1, 3)
    fa = np.asarray(fa); fb = np.asarray(fb)
    F = len(fa)
    binary = fb >= 0
    pa = lp[fa]; pb = lp[np.where(binary, fb, 0)]
    ca, sa = np.cos(pa[:, 2]), np.sin(pa[:, 2])
    dx, dy = pb[:, 0] - pa[:, 0], pb[:, 1] - pa[:, 1]
    Ja = np.zeros((F, 3, 3)); Jb = np.zeros((F, 3, 3)); r = np.zeros((F, 3))
    # xyt (april_graph_xyt.c:62-124): zhat = pa^-1 o pb
    Ja[:, 0, 0] = -ca; Ja[:, 0, 1] = -sa; Ja[:, 0, 2] = -sa * dx + ca * dy
    Ja[:, 1, 0] = sa;  Ja[:, 1, 1] = -ca; Ja[:, 1, 2] = -ca * dx - sa * dy
    Ja[:, 2, 2] = -1
    Jb[:, 0, 0] = ca;  Jb[:, 0, 1] = sa
    Jb[:, 1, 0] = -sa; Jb[:, 1, 1] = ca
    Jb[:, 2, 2] = 1
    r[:, 0] = z[:, 0] - (ca * dx + sa * dy)
    r[:, 1] = z[:, 1] - (-sa * dx + ca * dy)
    r[:, 2] = mod2pi(z[:, 2] - (pb[:, 2] - pa[:, 2]))
    # xytpos (april_graph_xytpos.c:63-102): J = I, r = z - p
    u = ~binary
    if u.any():
        Ja[u] = np.eye(3); Jb[u] = 0
        r[u, 0] = z[u, 0] - pa[u, 0]; r[u, 1] = z[u, 1] - pa[u, 1]; r[u, 2] = mod2pi(z[u, 2] - pa[u, 2])
    return Ja, Jb, r


def normal_equation_residual(lp, fa, fb, z, W, dx, lam=0.0, lam_nodes=None, chunk=1 << 20):
    """-> dict(max_abs_res, max_abs_rhs, max_rhs_terms, rel_max, rel_l2): residual of (J'WJ + lam I) dx = J'W r at lp, relative to
    the size of the right-hand side's terms (max over components of sum_f |J_f' W_f r_f|).
    lam_nodes: number of leading nodes that carry the Tikhonov term (all by default; the incremental path puts it only on the
    poses present at the last batch step, aprilsam.c:197-204 vs :508-542).  Works through the factors in chunks (memory)."""
    lp = np.asarray(lp, float).reshape(-1, 3); dx = np.asarray(dx, float).reshape(-1, 3)
    fa = np.asarray(fa, np.int64); fb = np.asarray(fb, np.int64)
    z = np.asarray(z, float).reshape(-1, 3); W = np.asarray(W, float).reshape(-1, 3, 3)
    N, F = len(lp), len(fa)
    res = np.zeros((N, 3)); rhs = np.zeros((N, 3)); scale = np.zeros((N, 3))
    for f0 in range(0, F, chunk):
        s = slice(f0, min(F, f0 + chunk))
        a, b = fa[s], fb[s]
        binary = b >= 0
        bb = np.where(binary, b, 0)
        Ja, Jb, r = linearise(lp, a, b, z[s])
        e = np.einsum("nij,nj->ni", Ja, dx[a]) + np.einsum("nij,nj->ni", Jb, dx[bb]) - r          # J dx - r
        We = np.einsum("nij,nj->ni", W[s], e); Wr = np.einsum("nij,nj->ni", W[s], r)
        ga = np.einsum("nji,nj->ni", Ja, We); gb = np.einsum("nji,nj->ni", Jb, We)
        ha = np.einsum("nji,nj->ni", Ja, Wr); hb = np.einsum("nji,nj->ni", Jb, Wr)
        for k in range(3):
            res[:, k] += np.bincount(a, weights=ga[:, k], minlength=N) + np.bincount(bb, weights=gb[:, k] * binary, minlength=N)
            rhs[:, k] += np.bincount(a, weights=ha[:, k], minlength=N) + np.bincount(bb, weights=hb[:, k] * binary, minlength=N)
            scale[:, k] += np.bincount(a, weights=np.abs(ha[:, k]), minlength=N) + np.bincount(bb, weights=np.abs(hb[:, k]) * binary, minlength=N)
    if lam > 0:
        n = N if lam_nodes is None else lam_nodes
        res[:n] += lam * dx[:n]
    # scale: sum_f |J_f' W_f r_f| per component, the terms of the right-hand side WITHOUT their cancellation -- near convergence J'W r
    # itself tends to zero while the rounding of its evaluation (here, and in any solver) stays at eps times these terms
    mr, mb, ms = float(np.max(np.abs(res))), float(np.max(np.abs(rhs))), float(np.max(scale))
    return dict(max_abs_res=mr, max_abs_rhs=mb, max_rhs_terms=ms, rel_max=mr / ms if ms > 0 else mr,
                rel_l2=float(np.linalg.norm(res) / max(np.linalg.norm(scale), 1e-300)))
