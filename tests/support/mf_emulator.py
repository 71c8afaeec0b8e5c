"""numpy emulation of the DEVICE algorithm (kernels.hip.h) driven by the product's symbolic plan.

TEST-ONLY.  It consumes exactly the arrays the GPU kernels consume (aprilsam_amd_plan_query): gather lists,
child->parent scatter maps, front layout with the right-hand side as an extra row, level order — so a
mismatch between the plan and what the kernels assume shows up on the CPU, without a GPU.  The numerics are
numpy's; nothing here is used by the product.
"""
import ctypes as C

import numpy as np


class PlanView:
    NAMES = ["perm", "pos", "front_first", "front_nsb", "front_nub", "front_parent", "front_level", "front_rows_ptr",
             "front_rows", "front_rel", "front_off", "ch_ptr", "ch_idx", "factor_front", "factor_la", "factor_lb",
             "factor_swap", "bd_front_ptr", "bd_row", "bd_col", "bd_src_ptr", "bd_src", "rd_front_ptr", "rd_col",
             "rd_src_ptr", "rd_src", "lev_ptr", "lev_fronts", "stats"]

    def __init__(self, lib, n_nodes, fa, fb, xy=None, leaf_nodes=16, shard_worlds=()):
        fn = np.ascontiguousarray(np.column_stack([fa, fb]).astype(np.int32))
        xyp = None
        if xy is not None:
            xy = np.ascontiguousarray(xy, float)
            xyp = xy.ctypes.data_as(C.POINTER(C.c_double))
        h = lib.dll.aprilsam_amd_plan_create(n_nodes, len(fa), fn.ctypes.data_as(C.POINTER(C.c_int)), xyp, leaf_nodes)
        try:
            for name in self.NAMES:
                out = C.POINTER(C.c_longlong)()
                n = lib.dll.aprilsam_amd_plan_query(h, name.encode(), C.byref(out))
                assert n >= 0, name
                setattr(self, name, np.array(out[:n], dtype=np.int64))
                lib.dll.aprilsam_amd_free(out)
            # ownership / exchange lists of sharded runs of this plan (aprilsam_amd_shard_plan), per requested world size
            self.shard = {}
            if shard_worlds:
                lib.dll.aprilsam_amd_shard_plan.restype = C.c_longlong
                lib.dll.aprilsam_amd_shard_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.c_longlong]
            for w in shard_worlds:
                lists = []
                for what in (1, 2, 3):
                    n = lib.dll.aprilsam_amd_shard_plan(h, w, what, None, 0)
                    buf = np.zeros(max(int(n), 1), np.int64)
                    lib.dll.aprilsam_amd_shard_plan(h, w, what, buf.ctypes.data_as(C.POINTER(C.c_longlong)), n)
                    lists.append(buf[:n])
                self.shard[w] = (lists[0].reshape(-1, 6), lists[1].reshape(-1, 5), lists[2])
        finally:
            lib.dll.aprilsam_amd_plan_destroy(h)
        self.N = n_nodes
        self.nF = int(self.stats[0]); self.nLevels = int(self.stats[1])


def contributions(oracle, lp, st_unary, fa, fb, z, W, swap):
    """what k_linearize writes: Hblk[3F,3,3] (final orientation), G[2F,3]"""
    F = len(fa)
    H = np.zeros((3 * F, 3, 3)); G = np.zeros((2 * F, 3))
    for f in range(F):
        binary = fb[f] >= 0
        J0, J1, r, _ = oracle.factor_eval(lp[fa[f]] if binary else st_unary[fa[f]], lp[fb[f]] if binary else None, z[f], W[f])
        J0 = J0.reshape(3, 3); J1 = J1.reshape(3, 3); Wm = np.asarray(W[f]).reshape(3, 3)
        JtW0 = J0.T @ Wm
        Haa = JtW0 @ J0
        H[3 * f] = np.triu(Haa) + np.triu(Haa, 1).T
        G[2 * f] = JtW0 @ r
        if binary:
            Hab = JtW0 @ J1
            H[3 * f + 1] = Hab.T if swap[f] else Hab
            JtW1 = J1.T @ Wm
            Hbb = JtW1 @ J1
            H[3 * f + 2] = np.triu(Hbb) + np.triu(Hbb, 1).T
            G[2 * f + 1] = JtW1 @ r
    return H, G


def solve(P, H, G, lam_pos):
    """multifrontal factorisation + back substitution exactly as the kernels do it; returns x[3N] by position"""
    nF = P.nF
    fronts = [None] * nF
    for t in range(nF):
        nsb, nub = int(P.front_nsb[t]), int(P.front_nub[t])
        nbc = nsb + nub
        R, Cc, ns = 3 * (nbc + 1), 3 * nbc, 3 * nsb
        Fm = np.zeros((R, Cc))
        first = int(P.front_first[t])
        for k in range(nsb):
            for d in range(3):
                Fm[3 * k + d, 3 * k + d] = lam_pos[first + k]
        for d in range(int(P.bd_front_ptr[t]), int(P.bd_front_ptr[t + 1])):
            br, bc = int(P.bd_row[d]), int(P.bd_col[d])
            assert br >= bc
            acc = np.zeros((3, 3))
            for s in range(int(P.bd_src_ptr[d]), int(P.bd_src_ptr[d + 1])):
                acc += H[int(P.bd_src[s])]
            if br == bc:
                acc = np.tril(acc)
            Fm[3 * br:3 * br + 3, 3 * bc:3 * bc + 3] += acc
        for d in range(int(P.rd_front_ptr[t]), int(P.rd_front_ptr[t + 1])):
            bc = int(P.rd_col[d])
            for s in range(int(P.rd_src_ptr[d]), int(P.rd_src_ptr[d + 1])):
                Fm[3 * nbc, 3 * bc:3 * bc + 3] += G[int(P.rd_src[s])]
        for ci in range(int(P.ch_ptr[t]), int(P.ch_ptr[t + 1])):
            c = int(P.ch_idx[ci])
            cns, cnu = int(P.front_nsb[c]), int(P.front_nub[c])
            U = fronts[c]
            rel = P.front_rel[int(P.front_rows_ptr[c]):int(P.front_rows_ptr[c + 1])]
            assert np.all(np.diff(rel) > 0)
            # scalar map of the child's update rows (+ rhs row) into this front
            m = np.concatenate([np.repeat(3 * rel, 3) + np.tile(np.arange(3), cnu), [3 * nbc]]).astype(int)
            src = np.concatenate([np.arange(3 * cns, 3 * (cns + cnu)), [3 * (cns + cnu)]])
            sub = np.tril(U[np.ix_(src, src[:-1])])
            Fm[np.ix_(m, m[:-1])] += sub
        # factor the first ns columns (lower-triangle storage)
        A11 = np.tril(Fm[:ns, :ns]); A11 = A11 + np.tril(A11, -1).T
        L11 = np.linalg.cholesky(A11)
        Fm[:ns, :ns] = L11
        L21 = np.linalg.solve(L11, Fm[ns:, :ns].T).T
        Fm[ns:, :ns] = L21
        upd = L21 @ L21[:Cc - ns].T
        Fm[ns:, ns:] -= upd
        fronts[t] = Fm
    x = np.zeros(3 * P.N)
    for l in range(P.nLevels - 1, -1, -1):
        for t in P.lev_fronts[int(P.lev_ptr[l]):int(P.lev_ptr[l + 1])]:
            t = int(t)
            nsb, nub = int(P.front_nsb[t]), int(P.front_nub[t])
            nbc = nsb + nub; ns = 3 * nsb
            Fm = fronts[t]
            rows = P.front_rows[int(P.front_rows_ptr[t]):int(P.front_rows_ptr[t + 1])]
            xs = x[(np.repeat(3 * rows, 3) + np.tile(np.arange(3), nub)).astype(int)]
            w = Fm[3 * nbc, :ns] - Fm[ns:3 * nbc, :ns].T @ xs
            xt = np.linalg.solve(Fm[:ns, :ns].T, w)
            first = int(P.front_first[t])
            x[3 * first:3 * first + ns] = xt
    return x
