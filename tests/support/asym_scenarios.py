"""Inputs whose information matrices are NOT symmetric as given -- what the reference's text loader produces for an EDGE2 line
with correlated information (examples/aprilsam_demo.c:73-75: I12, I13, I23 -> W[1], W[2], W[5]; W[3], W[6], W[7] stay zero).
The reference uses W as given and accumulates only the upper triangle of its ORDERED normal equations (aprilsam.c:162,171,520),
so which off-diagonal block a factor contributes depends on the reference's own elimination order.

TEST INFRASTRUCTURE: oracle/gen_golden.py --asym runs the unmodified reference on these and stores inputs + outputs under
tests/golden/asym_*.npz; the tests only read the fixtures.

The graphs carry xytpos priors on every few poses: with W used as given the reference's matrix is J'SJ + (a symmetric term of
the size of the correlations x the lever arm between the poses), and a pose graph held by ONE prior has eigenvalues of 1e-4
(M3500) that such a term drives negative -- the reference then walks into the NULL factor cs_chol returns (aprilsam.c:234-236).  Its INCREMENTAL factorisation does not even
notice: smatd_chol_inc_tr takes the square root of a negative pivot, every delta becomes NaN, xyt_node_update skips every pose
(april_graph_xyt.c:302-314) and the run carries on with frozen states -- a first version of the growth scenario (a prior on every fifth
pose) did exactly that from its eighth pose on, where this library reports "not positive definite" and recovers at the next step.
gen_golden.py therefore asserts that no step of the reference run produced a NaN delta: the fixture is a well-posed problem.
"""
import numpy as np

from aprilsam_amd import datasets


def batch_graph(n=300, extra=200, seed=11, spread=4.0, corr=0.05, prior_every=4):
    st, fa, fb, z, W = datasets.random_pose_graph(n, extra, seed, spread=spread)
    rng = np.random.default_rng(seed + 100)
    W = W.copy(); binary = fb >= 0
    W[binary] = datasets.as_loaded_with_correlations(np.tile(np.diag([60.0, 60.0, 150.0]).reshape(9), (int(binary.sum()), 1)), seed + 1, corr=corr)
    nodes = np.arange(0, n, prior_every, dtype=np.int32)
    pz = st[nodes] + rng.normal(0, 0.05, (len(nodes), 3))
    pW = datasets.as_loaded_with_correlations(np.tile(np.diag([40.0, 40.0, 80.0]).reshape(9), (len(nodes), 1)), seed + 2, corr=corr)
    return (st, np.concatenate([fa, nodes]).astype(np.int32), np.concatenate([fb, -np.ones(len(nodes), np.int32)]).astype(np.int32),
            np.vstack([z, pz]), np.vstack([W, pW]))


def growth_graph(n=800, corr=0.05, prior_every=2):
    """the first n poses of M3500 (loop closures included), its diagonal W given loader-style correlations, for harness.run_demo"""
    s0, a0, b0, z0, W0 = datasets.m3500_arrays()
    keep = np.maximum(a0, b0) < n
    nodes = np.arange(3, n, prior_every, dtype=np.int32)
    rng = np.random.default_rng(5)
    pz = s0[nodes] + rng.normal(0, 0.05, (len(nodes), 3))
    pW = datasets.as_loaded_with_correlations(np.tile(np.diag([8.0, 8.0, 20.0]).reshape(9), (len(nodes), 1)), 24, corr=corr)
    return (s0[:n], np.concatenate([a0[keep], nodes]).astype(np.int32), np.concatenate([b0[keep], -np.ones(len(nodes), np.int32)]).astype(np.int32),
            np.vstack([z0[keep], pz]), np.vstack([datasets.as_loaded_with_correlations(W0[keep], 22, corr=corr), pW]))


GROWTH_BATCH_EVERY = 250        # harness.run_demo(batch_every=...): a batch step in the middle of the incremental run re-orients every factor
