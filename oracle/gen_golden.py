#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference here.

TEST INFRASTRUCTURE.  Needs /root/reference (for data/M3500.txt and to build
oracle/_ref/libaprilsam_ref.so via `make -C oracle ref`); the fixtures it writes are plain numeric
arrays (inputs + expected outputs) so the tests can run where the reference tree does not exist.

    python oracle/gen_golden.py            # everything except the slow 100k lattice
    python oracle/gen_golden.py --big      # also K=316 (config 4), ~2 minutes of reference CPU time
    python oracle/gen_golden.py --custom   # ONLY the foreign-factor scenario (tests/support/custom_scenario.py)
    python oracle/gen_golden.py --asym     # ONLY the fixtures with information matrices that are not symmetric as given
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libaprilsam_ref.so")
REFDATA = "/root/reference/data/M3500.txt"


def batch_trace(ref, arrays, iters, keep_first=True):
    g = ref.new_graph(); g.build_from_arrays(*arrays)
    p = ref.new_param()
    chi2 = [g.chi2()]
    first = None
    for it in range(iters):
        g.cholesky(p)
        chi2.append(g.chi2())
        if it == 0:
            first = (g.states(), g.deltas())
    out = dict(chi2=np.array(chi2), final_states=g.states())
    if keep_first:
        out.update(states_after_1=first[0], dx_1=first[1])
    out["ref_nnzU"] = np.array([ref.dll.rs_param_U_nnz(p.ptr)], np.int64)
    ref.dll.rs_param_U_sumsq.restype = __import__("ctypes").c_double
    out["ref_sumsq"] = np.array([ref.dll.rs_param_U_sumsq(p.ptr)])
    p.destroy(); g.destroy()
    return out


def ref_ordering(ref, p, n):
    import ctypes as C
    order = np.zeros(n, np.int32)
    ref.dll.rs_param_ordering.argtypes = [C.c_void_p, C.c_void_p]
    ref.dll.rs_param_ordering(p.ptr, order.ctypes.data)
    return order


def asym_fixtures(ref):
    """Information matrices as the reference's loader leaves them when the file holds correlated information
    (examples/aprilsam_demo.c:73-75: upper triangle filled, lower left zero).  The reference uses W as given and accumulates only the
    upper triangle of its ORDERED matrix (aprilsam.c:171,520), so its normal equations depend on its own elimination order here:
    the fixtures hold that order too (the oracle takes it as data)."""
    from tests.support import asym_scenarios
    # (a) batch: a seeded random pose graph with priors, 3 iterations
    arr = asym_scenarios.batch_graph()
    st, fa, fb, z, W = arr
    g = ref.new_graph(); g.build_from_arrays(*arr); p = ref.new_param()
    chi2 = [g.chi2()]; states = []; dxs = []
    for _ in range(3):
        g.cholesky(p); chi2.append(g.chi2()); states.append(g.states()); dxs.append(g.deltas())
    order = ref_ordering(ref, p, len(st))
    np.savez_compressed(os.path.join(GOLD, "asym_batch.npz"), states=st, fa=fa, fb=fb, z=z, W=W, chi2=np.array(chi2), ordering=order,
                        states_after=np.array(states), dx=np.array(dxs))
    p.destroy(); g.destroy()
    print("asym batch: chi2", chi2)
    # (b) incremental growth with loop closures and a batch step every 250 poses
    arr = asym_scenarios.growth_graph()
    nan_steps = []

    def watch(k, p, wb):          # a NaN delta = the reference took the square root of a negative pivot: not a problem to pin anything against
        if np.isnan(watch.g.deltas()).any():
            nan_steps.append(k)
    G_ = type(ref.new_graph())
    orig_init = G_.__init__

    def grab(self, *a, **kw):
        orig_init(self, *a, **kw); watch.g = self
    G_.__init__ = grab
    try:
        res = harness.run_demo(ref, arr, deterministic=True, record_states_every=100, batch_every=asym_scenarios.GROWTH_BATCH_EVERY, on_step=watch)
    finally:
        G_.__init__ = orig_init
    assert not nan_steps, f"the reference produced NaN deltas at steps {nan_steps[:10]}: ill-posed scenario"
    np.savez_compressed(os.path.join(GOLD, "asym_inc_demo.npz"), states=arr[0], fa=arr[1], fb=arr[2], z=arr[3], W=arr[4], chi2=res["chi2"],
                        was_batch=res["was_batch"], final_states=res["final_states"],
                        **{f"snap_{k}": v for k, v in res["snaps"].items()})
    print("asym inc demo: final chi2", res["chi2"][-1], "batch steps", int(res["was_batch"].sum()) - 1, "loop closures",
          int(((np.abs(arr[1] - arr[2]) > 1) & (arr[2] >= 0)).sum()))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--big", action="store_true"); ap.add_argument("--custom", action="store_true")
    ap.add_argument("--asym", action="store_true")
    a = ap.parse_args()
    import ctypes as C
    os.makedirs(GOLD, exist_ok=True)
    ref = host.SolverLib(REFLIB)
    ref.dll.rs_param_U_nnz.restype = C.c_longlong
    if a.custom:      # factor types the reference only knows through their vtable
        import tempfile
        from tests.support import custom_scenario
        cl = custom_scenario.build_custom_lib(tempfile.mkdtemp())
        out = custom_scenario.run(ref, cl)
        np.savez_compressed(os.path.join(GOLD, "custom_factors.npz"), **out)
        print("custom factors: chi2", out["chi2"])
        out3 = custom_scenario.run(ref, cl, triples=8)          # ... with three-pose factors (factor->nnodes == 3)
        np.savez_compressed(os.path.join(GOLD, "custom_factors3.npz"), **out3)
        print("custom factors incl. three-pose ones: chi2", out3["chi2"])
        return
    if a.asym:
        asym_fixtures(ref)
        return
    prod = host.SolverLib()          # only its data generators are used here (lattice arrays)

    # 1. M3500 input, parsed once from the reference's data file (data fixture)
    st, fa, fb, z, W = datasets.parse_vertex_edge_text(REFDATA)
    np.savez_compressed(os.path.join(GOLD, "m3500_input.npz"), states=st, fa=fa, fb=fb, z=z, W=W)
    print("M3500:", st.shape, fa.shape)

    # 2. M3500 batch, 10 iterations (config 1)
    out = batch_trace(ref, datasets.with_prior(st, fa, fb, z, W, first=True), 10)
    np.savez_compressed(os.path.join(GOLD, "m3500_batch.npz"), **out)
    print("M3500 chi2:", out["chi2"])

    # 3. per-factor evaluation on seeded poses, incl. theta at the wrap edges
    rng = np.random.default_rng(12345)
    n = 400
    pa = np.column_stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-np.pi, np.pi, n)])
    pb = np.column_stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-np.pi, np.pi, n)])
    zz = np.column_stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-np.pi, np.pi, n)])
    edge = [np.pi, -np.pi, np.pi - 1e-12, -np.pi + 1e-12, 0.0, 3 * np.pi, -3 * np.pi, 2 * np.pi]
    for k, e in enumerate(edge):      # residual angle lands on / next to the wrap points
        pa[k, 2] = 0.3; pb[k, 2] = 0.3 + 0.1; zz[k, 2] = 0.1 + e
    Wf = np.empty((n, 9))
    for k in range(n):
        M = rng.normal(size=(3, 3)); Wk = M @ M.T + np.eye(3); Wf[k] = ((Wk + Wk.T) / 2).reshape(9)
    g = ref.new_graph()
    states = np.vstack([pa, pb])
    fa_e = np.arange(n, dtype=np.int32); fb_e = np.arange(n, 2 * n, dtype=np.int32)
    # second half: xytpos factors on the pa nodes
    g.build_from_arrays(states, np.concatenate([fa_e, fa_e]), np.concatenate([fb_e, -np.ones(n, np.int32)]),
                        np.vstack([zz, zz]), np.vstack([Wf, Wf]))
    J0 = np.zeros((2 * n, 9)); J1 = np.zeros((2 * n, 9)); r = np.zeros((2 * n, 3)); chi2 = np.zeros(2 * n)
    dp = C.POINTER(C.c_double)
    for f in range(2 * n):
        j0 = (C.c_double * 9)(); j1 = (C.c_double * 9)(); rr = (C.c_double * 3)(); ww = (C.c_double * 9)(); c2 = C.c_double()
        ref.dll.rs_factor_eval(g.ptr, f, 0, j0, j1, rr, ww, C.byref(c2))
        J0[f] = list(j0); J1[f] = list(j1); r[f] = list(rr); chi2[f] = c2.value
    np.savez_compressed(os.path.join(GOLD, "factor_eval.npz"), pa=pa, pb=pb, z=zz, W=Wf, J0=J0, J1=J1, r=r, chi2=chi2,
                        graph_chi2=np.array([g.chi2()]))
    g.destroy()

    # 4. tutorial, both modes
    for mode, name in ((True, "batch"), (False, "inc")):
        res = harness.run_tutorial(ref, batch_update_only=mode)
        np.savez_compressed(os.path.join(GOLD, f"tutorial_{name}.npz"), chi2=np.array([c for c, _ in res]),
                            **{f"states_{i}": s for i, (_, s) in enumerate(res)})
        print("tutorial", name, [round(c, 6) for c, _ in res])

    # 4b. incremental demo (examples/aprilsam_demo.c semantics), deterministic schedule: the wall-clock rule
    #     aprilsam.c:557 is neutralised by setting param->batch_time = 1e300 before every incremental call
    res = harness.run_demo(ref, (st, fa, fb, z, W), deterministic=True)
    np.savez_compressed(os.path.join(GOLD, "m3500_inc_demo.npz"), chi2=res["chi2"], was_batch=res["was_batch"],
                        final_states=res["final_states"], ref_ms=res["ms"])
    print("inc demo: final chi2", res["chi2"][-1], "fallbacks", int(res["was_batch"].sum()) - 1,
          "at node counts", (np.nonzero(res["was_batch"])[0] + 1)[:8], "total ms", res["ms"].sum())

    # 5. lattices
    for K, iters in ((6, 4), (24, 4), (60, 3)) + (((120, 2),) if a.big else ()):
        out = batch_trace(ref, prod.lattice_arrays(K), iters)
        np.savez_compressed(os.path.join(GOLD, f"lattice_{K}.npz"), **out)
        print("lattice", K, out["chi2"], "ref nnzU", out["ref_nnzU"], out["ref_sumsq"])
    if a.big:
        out = batch_trace(ref, prod.lattice_arrays(316), 1, keep_first=False)
        np.savez_compressed(os.path.join(GOLD, "lattice_316.npz"), chi2=out["chi2"], ref_nnzU=out["ref_nnzU"],
                            ref_sumsq=out["ref_sumsq"], states_sample=out["final_states"][::997])
        print("lattice 316", out["chi2"], out["ref_nnzU"], out["ref_sumsq"])

    # 6. seeded random pose graphs with full information matrices
    for seed, (n, extra) in enumerate(((12, 6), (80, 60), (400, 350), (1500, 900))):
        arr = datasets.random_pose_graph(n, extra, seed)
        out = batch_trace(ref, arr, 3)
        np.savez_compressed(os.path.join(GOLD, f"random_{seed}.npz"), **out)
        print("random", seed, out["chi2"])

    # 7. normal equations of a small graph after one batch call (A upper, B), un-permuted to node coords
    arr = datasets.random_pose_graph(10, 5, 7)
    g = ref.new_graph(); g.build_from_arrays(*arr); p = ref.new_param()
    g.cholesky(p)
    nA = ref.dll.rs_param_A_nnz(p.ptr); ncol = ref.dll.rs_param_n(p.ptr)
    ri = np.zeros(nA, np.int32); ci = np.zeros(nA, np.int32); v = np.zeros(nA); B = np.zeros(ncol); order = np.zeros(10, np.int32)
    ip = C.POINTER(C.c_int)
    ref.dll.rs_param_A_dump(p.ptr, ri.ctypes.data_as(ip), ci.ctypes.data_as(ip), v.ctypes.data_as(dp))
    ref.dll.rs_param_B(p.ptr, B.ctypes.data_as(dp)); ref.dll.rs_param_ordering(p.ptr, order.ctypes.data_as(ip))
    # scalar index 3*position+d -> 3*node+d
    A = np.zeros((ncol, ncol))
    tonode = lambda s: 3 * order[s // 3] + s % 3   # noqa: E731
    Bn = np.zeros(ncol)
    for k in range(nA):
        i, j = tonode(ri[k]), tonode(ci[k])
        A[min(i, j), max(i, j)] += v[k] if i <= j else 0
        if i > j:
            A[j, i] += v[k]
    for s_ in range(ncol):
        Bn[tonode(s_)] = B[s_]
    np.savez_compressed(os.path.join(GOLD, "normal_eq_10.npz"), A_upper=A, B=Bn, lp=arr[0])
    p.destroy(); g.destroy()
    print("done")


if __name__ == "__main__":
    main()
