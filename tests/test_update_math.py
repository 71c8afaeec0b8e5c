"""The formulas behind the incremental path's low-rank front update (kernels.hip.h: front_update_body), replayed in numpy on
a two-front assembly tree and compared with a fresh factorisation: per front P = L11^-1 W1, the per-column scalars from the
prefix sums T_j = I + sum_{l<j} p_l p_l^T, the row recurrence, W~ = (W2 - L21 P) C^-T handed to the parent through the block
map, S' = S + W~ W~^T -- including a structure that GROWS (the new pose becomes a row of the child front: zero rows of L in a
fresh layout) and two factors applied one after the other.  No GPU involved: this pins the algebra the kernel implements."""
import numpy as np


def _update_front(L, S, W):
    """L: (rows x ns) panel [L11; L21], S: Schur update over the rows below, W: (rows x 3) incoming vectors.
    Returns L', S', W~ (rows below x 3) exactly as front_update_body forms them."""
    n, ns = L.shape
    K = W.shape[1]
    P = np.zeros((ns, K)); r = W[:ns].copy()
    for j in range(ns):                                   # forward substitution (the v_readlane chain)
        P[j] = r[j] / L[j, j]
        r[j + 1:] -= np.outer(L[j + 1:ns, j], P[j])
    T = np.eye(K); U = np.zeros((ns, K)); g = np.zeros(ns)
    for j in range(ns):                                   # (a wave scan on the device)
        U[j] = np.linalg.solve(T, P[j]); g[j] = np.sqrt(1 + P[j] @ U[j]); T = T + np.outer(P[j], P[j])
    Ln = L.copy(); Rr = W.copy()
    for i in range(n):                                    # every row on its own
        rr = W[i].copy()
        for j in range(min(i + 1, ns)):
            rr = rr - L[i, j] * P[j]
            Ln[i, j] = g[j] * L[i, j] + (rr @ U[j]) / g[j]
        Rr[i] = rr
    C = np.linalg.cholesky(T)
    Wt = np.linalg.solve(C, Rr[ns:].T).T
    return Ln, S + Wt @ Wt.T, Wt


def test_rank3_updates_through_two_fronts_equal_a_fresh_factorisation():
    rng = np.random.default_rng(3)
    nc, npar = 9, 12                                      # child own columns; parent own columns (the last 3: the "new pose")
    n = nc + npar
    crows = np.array([0, 1, 2, 5, 6, 7])                  # child's update rows before the step, as rows of the parent
    M = rng.normal(size=(n + 1, n + 1)) * 0.2             # (+1: the right-hand-side row rides along as a row of the matrix)
    A = np.zeros((n + 1, n + 1))
    A[:nc, :nc] = M[:nc, :nc] @ M[:nc, :nc].T + 3 * np.eye(nc)
    A[nc:n, nc:n] = M[nc:n, nc:n] @ M[nc:n, nc:n].T + 3 * np.eye(npar)
    cpl = rng.normal(size=(len(crows), nc)) * 0.3
    A[nc + crows[:, None], np.arange(nc)[None, :]] = cpl; A[:nc, nc + crows] = cpl.T
    A[n, :n] = rng.normal(size=n); A[:n, n] = A[n, :n]; A[n, n] = 50.0

    def fresh(Am):
        Lf = np.linalg.cholesky(Am + 0)
        return Lf
    # fronts before the step
    Lf = fresh(A)
    rows_c = np.concatenate([np.arange(nc), nc + crows, [n]])            # child: own, update rows, rhs row
    Lc = Lf[np.ix_(rows_c, np.arange(nc))]
    Sc = -Lc[nc:] @ Lc[nc:].T                                            # the child's contribution to its parent (nothing else assembled there)
    rows_p = np.concatenate([np.arange(nc, n), [n]])
    Lp = Lf[np.ix_(rows_p, np.arange(nc, n))]
    # two new factors: (child pose block 1, new pose = parent's last block) and (child pose block 2, parent block 0)
    V = np.zeros((n + 1, 6))
    V[3:6, 0:3] = rng.normal(size=(3, 3)); V[n - 3:n, 0:3] = rng.normal(size=(3, 3)); V[n, 0:3] = rng.normal(size=3)
    V[6:9, 3:6] = rng.normal(size=(3, 3)); V[nc:nc + 3, 3:6] = rng.normal(size=(3, 3)); V[n, 3:6] = rng.normal(size=3)
    A2 = A + V @ V.T
    L2 = fresh(A2)
    # child: its structure gains the new pose's rows (zero rows of L, zero rows / columns of S), appended before the rhs row
    new_rows = np.arange(n - 3, n)
    rows_c2 = np.concatenate([np.arange(nc), nc + crows, new_rows, [n]])
    k = len(crows)
    Lc2 = np.zeros((len(rows_c2), nc)); Lc2[:nc + k] = Lc[:nc + k]; Lc2[-1] = Lc[-1]
    Sc2 = np.zeros((k + 3 + 1, k + 3 + 1)); idx = np.r_[np.arange(k), k + 3]
    Sc2[np.ix_(idx, idx)] = Sc
    Wt_all = []
    for f in range(2):                                     # factor after factor
        Lc2, Sc2, Wt = _update_front(Lc2, Sc2, V[rows_c2][:, 3 * f:3 * f + 3])
        Wt_all.append(Wt)
    assert np.max(np.abs(Lc2 - L2[np.ix_(rows_c2, np.arange(nc))])) < 1e-12
    V2 = V[rows_c2[nc:]]
    Sref = V2 @ V2.T - Lc2[nc:] @ Lc2[nc:].T                             # the new factors are assembled in the child (it owns them)
    assert np.max(np.abs(Sc2 - Sref)) < 1e-12
    # parent: the child's vectors arrive through its block map (update rows of the child -> rows of the parent); the parent owns
    # no new factor here except the parts of V in its own rows, which came along inside W~ (V's rows there are update rows of the child)
    for f in range(2):
        Wp = np.zeros((npar + 1, 3))
        Wp[rows_c2[nc:] - nc] = Wt_all[f]
        Lp, _, _ = _update_front(Lp, np.zeros((1, 1)), Wp)
    assert np.max(np.abs(Lp - L2[np.ix_(rows_p, np.arange(nc, n))])) < 1e-12
