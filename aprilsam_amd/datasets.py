"""Input data for the path's configurations (BASELINE.json `configs`), as plain arrays.

* M3500: the Manhattan-world pose graph the reference ships as data/M3500.txt.  `parse_vertex_edge_text`
  restates the reference loader (examples/aprilsam_demo.c:52-99): VERTEX2 id x y t -> node with
  state = init = truth; EDGE2 a b dx dy dt I11 I12 I22 I33 I13 I23 -> xyt factor with
  W[0]=I11, W[1]=I12, W[4]=I22, W[8]=I33, W[2]=I13, W[5]=I23 (lower half left zero, as the reference
  does).  The parsed arrays are committed as the fixture tests/golden/m3500_input.npz (the GPU box has
  no /root/reference), `m3500_arrays()` loads them.
* Batch configuration = all nodes + all factors + the harness prior on node 0 with
  W = diag(1e4, 1e4, 1e3), z = 0 (examples/aprilsam_demo.c:133-145); `with_prior()` appends it.
* The synthetic lattice comes from the product library itself
  (aprilsam_amd_lattice_arrays) — see host.SolverLib.lattice_arrays.
"""
import os

import numpy as np

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def parse_vertex_edge_text(path):
    """examples/aprilsam_demo.c:52-99 restated. Returns (states[N,3], fa[F], fb[F], z[F,3], W[F,9])."""
    states, fa, fb, z, W = [], [], [], [], []
    with open(path) as f:
        tok = f.read().split()
    i = 0
    while i < len(tok):
        if tok[i] == "VERTEX2":
            states.append([float(tok[i + 2]), float(tok[i + 3]), float(tok[i + 4])])
            i += 5
        elif tok[i] == "EDGE2":
            a, b = int(tok[i + 1]), int(tok[i + 2])
            v = [float(t) for t in tok[i + 3:i + 12]]
            w = [0.0] * 9
            w[0], w[1], w[4], w[8], w[2], w[5] = v[3], v[4], v[5], v[6], v[7], v[8]
            fa.append(a); fb.append(b); z.append(v[:3]); W.append(w)
            i += 12
        else:
            raise ValueError(f"unexpected token {tok[i]!r}")
    return (np.array(states, float), np.array(fa, np.int32), np.array(fb, np.int32),
            np.array(z, float).reshape(-1, 3), np.array(W, float).reshape(-1, 9))


def m3500_arrays():
    d = np.load(os.path.join(_GOLDEN, "m3500_input.npz"))
    return d["states"], d["fa"], d["fb"], d["z"], d["W"]


PRIOR_W = np.array([1e4, 0, 0, 0, 1e4, 0, 0, 0, 1e3])


def with_prior(states, fa, fb, z, W, first=True):
    """Append (or prepend, like the demo at pose 0) the xytpos prior on node 0; fb = -1 marks it."""
    pa, pb, pz, pW = np.array([0], np.int32), np.array([-1], np.int32), np.zeros((1, 3)), PRIOR_W.reshape(1, 9)
    if first:
        return states, np.concatenate([pa, fa]), np.concatenate([pb, fb]), np.vstack([pz, z]), np.vstack([pW, W])
    return states, np.concatenate([fa, pa]), np.concatenate([fb, pb]), np.vstack([z, pz]), np.vstack([W, pW])


def m3500_batch():
    """Config 1/2: full M3500 graph, prior first (the demo adds it with pose 0)."""
    return with_prior(*m3500_arrays(), first=True)


def random_pose_graph(n_nodes, extra_edges, seed, spread=10.0):
    """Seeded random connected pose graph for parity tests: odometry chain + random loop closures,
    full (symmetric positive definite) information matrices, angles over the whole circle."""
    rng = np.random.default_rng(seed)
    states = np.column_stack([rng.uniform(-spread, spread, n_nodes), rng.uniform(-spread, spread, n_nodes),
                              rng.uniform(-np.pi, np.pi, n_nodes)])
    pairs = [(i, i + 1) for i in range(n_nodes - 1)]
    seen = set(pairs)
    while len(pairs) < n_nodes - 1 + extra_edges:
        a, b = sorted(rng.integers(0, n_nodes, 2).tolist())
        if a != b and (a, b) not in seen:
            seen.add((a, b)); pairs.append((a, b))
    fa = np.array([p[0] for p in pairs], np.int32); fb = np.array([p[1] for p in pairs], np.int32)
    # flip the direction of some edges so both orientations occur
    flip = rng.random(len(pairs)) < 0.3
    fa, fb = np.where(flip, fb, fa).astype(np.int32), np.where(flip, fa, fb).astype(np.int32)
    F = len(pairs)
    z = np.empty((F, 3)); W = np.empty((F, 9))
    for k in range(F):
        pa, pb = states[fa[k]], states[fb[k]]
        c, s = np.cos(pa[2]), np.sin(pa[2])
        dx, dy = pb[0] - pa[0], pb[1] - pa[1]
        z[k] = [c * dx + s * dy + rng.normal(0, 0.3), -s * dx + c * dy + rng.normal(0, 0.3),
                pb[2] - pa[2] + rng.normal(0, 0.1)]
        M = rng.normal(size=(3, 3))
        Wk = M @ M.T + np.diag([20.0, 20.0, 50.0])
        W[k] = ((Wk + Wk.T) / 2).reshape(9)
    return with_prior(states, fa, fb, z, W, first=bool(seed % 2))


def write_vertex_edge_text(path, states, fa, fb, z, W):
    """inverse of parse_vertex_edge_text (lossless: 17 significant digits)"""
    with open(path, "w") as f:
        for i, s in enumerate(states):
            f.write("VERTEX2 %d %.17g %.17g %.17g\n" % (i, s[0], s[1], s[2]))
        for k in range(len(fa)):
            w = W[k]
            f.write("EDGE2 %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n" % (
                fa[k], fb[k], z[k][0], z[k][1], z[k][2], w[0], w[1], w[4], w[8], w[2], w[5]))


def as_loaded_with_correlations(W, seed, corr=0.04):
    """Information matrices as the reference's text loader leaves them for an EDGE2 line with correlated information
    (examples/aprilsam_demo.c:73-75): I12, I13, I23 go to W[1], W[2], W[5]; W[3], W[6], W[7] stay ZERO -- the matrix is not
    symmetric as given, and the reference uses it as given (aprilsam.c:162,171,520).  The correlations are
    `corr` x sqrt(W_ii W_jj) with random signs: small enough for the reference's system to stay positive definite (it walks
    into a NULL factor otherwise, aprilsam.c:234-236).  Rows with fb < 0 semantics (priors) are the caller's to leave alone."""
    rng = np.random.default_rng(seed)
    W = np.array(W, float).reshape(-1, 9).copy()
    d = np.sqrt(np.abs(W[:, [0, 4, 8]]))
    sg = rng.choice([-1.0, 1.0], size=(len(W), 3)) * rng.uniform(0.5, 1.0, size=(len(W), 3))
    W[:, 1] = corr * sg[:, 0] * d[:, 0] * d[:, 1]
    W[:, 2] = corr * sg[:, 1] * d[:, 0] * d[:, 2]
    W[:, 5] = corr * sg[:, 2] * d[:, 1] * d[:, 2]
    W[:, 3] = 0.0; W[:, 6] = 0.0; W[:, 7] = 0.0
    return W
