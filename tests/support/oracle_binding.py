"""ctypes binding of oracle/liboracle.so (the plain-C CPU restatement).  TEST-ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libaprilsam_ref.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class Oracle:
    def __init__(self, path=LIB):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run `make -C oracle liboracle.so` (or __graft_entry__.build())")
        self.dll = C.CDLL(path)
        d = self.dll
        d.orc_mod2pi.restype = C.c_double; d.orc_mod2pi.argtypes = [C.c_double]
        d.orc_factor_eval.restype = C.c_double
        d.orc_factor_eval.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        d.orc_chi2.restype = C.c_double
        d.orc_chi2.argtypes = [C.c_int, _dp, C.c_int, _ip, _ip, _dp, _dp]
        d.orc_solve_system.argtypes = [C.c_int, _dp, _dp, C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, _ip, _dp]
        d.orc_batch_step.argtypes = [C.c_int, _dp, C.c_int, _ip, _ip, _dp, _dp, C.c_double, _dp, _dp]
        d.orc_batch_step_ordered.argtypes = [C.c_int, _dp, C.c_int, _ip, _ip, _dp, _dp, C.c_double, _ip, _dp, _dp]
        d.orc_normal_equations_dense.argtypes = [C.c_int, _dp, _dp, C.c_int, _ip, _ip, _dp, _dp, C.c_double, _dp, _dp]

    @staticmethod
    def _prep(states, fa, fb, z, W):
        return (np.ascontiguousarray(states, float).copy(), np.ascontiguousarray(fa, np.int32),
                np.ascontiguousarray(fb, np.int32), np.ascontiguousarray(z, float),
                np.ascontiguousarray(W, float).reshape(-1, 9))

    def mod2pi(self, v):
        return self.dll.orc_mod2pi(float(v))

    def factor_eval(self, pa, pb, z, W):
        J0 = np.zeros(9); J1 = np.zeros(9); r = np.zeros(3)
        pa = np.ascontiguousarray(pa, float); z = np.ascontiguousarray(z, float); W = np.ascontiguousarray(W, float)
        if pb is None:
            c = self.dll.orc_factor_eval(0, _d(pa), None, _d(z), _d(W), _d(J0), _d(J1), _d(r))
        else:
            pb = np.ascontiguousarray(pb, float)
            c = self.dll.orc_factor_eval(1, _d(pa), _d(pb), _d(z), _d(W), _d(J0), _d(J1), _d(r))
        return J0, J1, r, c

    def chi2(self, states, fa, fb, z, W):
        s, fa, fb, z, W = self._prep(states, fa, fb, z, W)
        return self.dll.orc_chi2(len(s), _d(s), len(fa), _i(fa), _i(fb), _d(z), _d(W))

    def batch_step(self, states, fa, fb, z, W, lam=1e-4, order=None):
        """returns (new_states, dx, stats[nnzL, sumsq]); raises on non-SPD.  order: elimination order (position -> node) -- only matters,
        and is then required for parity with the reference, when some W is not symmetric as given (oracle.h)"""
        s, fa, fb, z, W = self._prep(states, fa, fb, z, W)
        dx = np.zeros_like(s); stats = np.zeros(2)
        if order is not None:
            order = np.ascontiguousarray(order, np.int32)
            assert len(order) == len(s)
        rc = self.dll.orc_batch_step_ordered(len(s), _d(s), len(fa), _i(fa), _i(fb), _d(z), _d(W), lam, _i(order) if order is not None else None, _d(dx), _d(stats))
        if rc != 0:
            raise ArithmeticError("oracle: matrix not positive definite")
        return s, dx, stats

    def solve_system(self, lp, st_unary, fa, fb, z, W, lambda_node):
        lp, fa, fb, z, W = self._prep(lp, fa, fb, z, W)
        su = np.ascontiguousarray(st_unary, float); lam = np.ascontiguousarray(lambda_node, float)
        dx = np.zeros_like(lp)
        rc = self.dll.orc_solve_system(len(lp), _d(lp), _d(su), len(fa), _i(fa), _i(fb), _d(z), _d(W), _d(lam), _d(dx), None, None)
        if rc != 0:
            raise ArithmeticError("oracle: matrix not positive definite")
        return dx

    def normal_equations(self, lp, fa, fb, z, W, lam=1e-4):
        lp, fa, fb, z, W = self._prep(lp, fa, fb, z, W)
        n = 3 * len(lp)
        A = np.zeros((n, n)); B = np.zeros(n)
        self.dll.orc_normal_equations_dense(len(lp), _d(lp), _d(lp), len(fa), _i(fa), _i(fb), _d(z), _d(W), lam, _d(A), _d(B))
        return A, B

    def iterate(self, arrays, iters, lam=1e-4, order=None):
        """chi2 trace [iters+1] and final states of `iters` batch steps"""
        s, fa, fb, z, W = self._prep(*arrays)
        chi2 = [self.chi2(s, fa, fb, z, W)]
        for _ in range(iters):
            s, _, _ = self.batch_step(s, fa, fb, z, W, lam, order)
            chi2.append(self.chi2(s, fa, fb, z, W))
        return np.array(chi2), s
