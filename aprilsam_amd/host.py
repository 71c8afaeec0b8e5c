"""Host-side Python mirror of the reference's graph / solver API for this path.

Same names and argument meaning as the C API (aprilsam/aprilsam.h:184-301): a `Graph` holds xyt
nodes and xyt / xytpos factors; `cholesky()` = april_graph_cholesky, `cholesky_inc()` =
april_graph_cholesky_inc, `chi2()` = april_graph_chi2.  Everything goes through the C-ABI of a
shared library that exports the reference's symbols — libaprilsam_amd.so (the product) by default.
The wrapper is ABI-generic on purpose: the parity tests load oracle/_ref/libaprilsam_ref.so (the
unmodified reference) through the very same class, so both sides of a comparison are driven by
identical Python code.

This module is plumbing only: no numerics happen in Python.
"""
import ctypes as C
import os
import os

import numpy as np

from . import abi

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# APRILSAM_AMD_LIB: another build of the same library (tools/sanitize_host.sh points it at the ASan / UBSan build of the host sources)
PRODUCT_LIB = os.environ.get("APRILSAM_AMD_LIB") or os.path.join(_PKG_DIR, "lib", "libaprilsam_amd.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _np_d(a):
    return a.ctypes.data_as(_dp)


def _np_i(a):
    return a.ctypes.data_as(_ip)


class SolverLib:
    """A loaded shared library exporting the reference API names."""

    def __init__(self, path=None):
        self.path = path or PRODUCT_LIB
        if not os.path.exists(self.path):
            raise FileNotFoundError(
                f"{self.path} not found — build it first: python -c 'import __graft_entry__ as g; g.build()'")
        self.dll = C.CDLL(self.path, mode=os.RTLD_LOCAL if hasattr(os, "RTLD_LOCAL") else 0)
        d = self.dll
        self.is_product = hasattr(d, "aprilsam_amd_version")
        d.april_graph_create.restype = C.POINTER(abi.Graph)
        d.april_graph_destroy.argtypes = [C.POINTER(abi.Graph)]
        d.april_graph_node_xyt_create.restype = C.POINTER(abi.Node)
        d.april_graph_node_xyt_create.argtypes = [_dp, _dp, _dp]
        d.april_graph_factor_xyt_create.restype = C.POINTER(abi.Factor)
        d.april_graph_factor_xyt_create.argtypes = [C.c_int, C.c_int, _dp, _dp, C.POINTER(abi.Matd3x3)]
        d.april_graph_factor_xytpos_create.restype = C.POINTER(abi.Factor)
        d.april_graph_factor_xytpos_create.argtypes = [C.c_int, _dp, _dp, C.POINTER(abi.Matd3x3)]
        d.april_graph_chi2.restype = C.c_double
        d.april_graph_chi2.argtypes = [C.POINTER(abi.Graph)]
        for name in ("april_graph_cholesky", "april_graph_cholesky_inc"):
            getattr(d, name).argtypes = [C.POINTER(abi.Graph), C.POINTER(abi.CholeskyParam)]
            getattr(d, name).restype = None
        d.april_graph_cholesky_inc_solver.argtypes = [C.POINTER(abi.Graph), C.POINTER(abi.CholeskyParam), _ip]
        d.april_graph_cholesky_inc_solver.restype = None
        d.april_graph_cholesky_param_init.argtypes = [C.POINTER(abi.CholeskyParam)]
        d.april_graph_cholesky_param_destory.argtypes = [C.POINTER(abi.CholeskyParam)]
        if self.is_product:
            self._add_node = d.aprilsam_amd_graph_add_node
            self._add_factor = d.aprilsam_amd_graph_add_factor
            d.aprilsam_amd_version.restype = C.c_char_p
            d.aprilsam_amd_get_stats.argtypes = [C.POINTER(abi.CholeskyParam), C.POINTER(abi.Stats)]
            d.aprilsam_amd_set_option.argtypes = [C.c_char_p, C.c_double]
            if hasattr(d, "aprilsam_amd_get_option"):          # (absent from builds of earlier rounds, which tools/ A/B against this one)
                d.aprilsam_amd_get_option.argtypes = [C.c_char_p, _dp]
                d.aprilsam_amd_debug_guard_selftest.argtypes = [C.POINTER(abi.CholeskyParam)]
            d.aprilsam_amd_last_error.argtypes = [C.c_char_p, C.c_int]
            d.aprilsam_amd_batch_resident.argtypes = [C.POINTER(abi.Graph), C.POINTER(abi.CholeskyParam),
                                                      C.c_int, _dp, _dp]
            for nm in ("begin", "sync", "end"):
                getattr(d, f"aprilsam_amd_resident_{nm}").argtypes = [C.POINTER(abi.Graph), C.POINTER(abi.CholeskyParam)]
            d.aprilsam_amd_resident_steps.argtypes = [C.POINTER(abi.Graph), C.POINTER(abi.CholeskyParam), C.c_int, C.c_int]
            d.aprilsam_amd_resident_chi2.argtypes = [C.POINTER(abi.Graph)]
            d.aprilsam_amd_resident_chi2.restype = C.c_double
            d.aprilsam_amd_set_device.argtypes = [C.c_int]
            if hasattr(d, "aprilsam_amd_param_set_device"):
                d.aprilsam_amd_param_set_device.argtypes = [C.POINTER(abi.CholeskyParam), C.c_int]
                d.aprilsam_amd_param_get_device.argtypes = [C.POINTER(abi.CholeskyParam)]
            d.aprilsam_amd_make_lattice.argtypes = [C.POINTER(abi.Graph), C.c_int]
            d.aprilsam_amd_lattice_arrays.argtypes = [C.c_int, _dp, _ip, _ip, _dp, _dp]
            d.aprilsam_amd_graph_from_arrays.argtypes = [C.POINTER(abi.Graph), C.c_int, _dp, C.c_int, _ip, _ip, _dp, _dp]
            if hasattr(d, "aprilsam_amd_graph_node_arrays"):
                d.aprilsam_amd_graph_node_arrays.argtypes = [C.POINTER(abi.Graph), _dp, _dp, _dp]
                d.aprilsam_amd_graph_node_arrays.restype = None
            d.aprilsam_amd_plan_create.restype = C.c_void_p
            d.aprilsam_amd_plan_create.argtypes = [C.c_int, C.c_int, _ip, _dp, C.c_int]
            d.aprilsam_amd_plan_destroy.argtypes = [C.c_void_p]
            d.aprilsam_amd_plan_query.restype = C.c_longlong
            d.aprilsam_amd_plan_query.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_longlong))]
            d.aprilsam_amd_free.argtypes = [C.c_void_p]
            d.aprilsam_amd_graph_save_ex.argtypes = [C.POINTER(abi.Graph), C.c_char_p, C.c_ulonglong]
            d.aprilsam_amd_graph_load.restype = C.POINTER(abi.Graph)
            d.aprilsam_amd_graph_load.argtypes = [C.c_char_p]
            d.aprilsam_amd_attr_put_string.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p]
            d.aprilsam_amd_attr_get_string.restype = C.c_char_p
            d.aprilsam_amd_attr_get_string.argtypes = [C.c_void_p, C.c_char_p]
        else:  # the reference + oracle/ref_shim.c helpers
            self._add_node = d.rs_graph_add_node
            self._add_factor = d.rs_graph_add_factor
        d.april_graph_save.argtypes = [C.POINTER(abi.Graph), C.c_char_p]
        d.april_graph_create_from_file.restype = C.POINTER(abi.Graph)
        d.april_graph_create_from_file.argtypes = [C.c_char_p]
        self._add_node.argtypes = [C.POINTER(abi.Graph), C.POINTER(abi.Node)]
        self._add_factor.argtypes = [C.POINTER(abi.Graph), C.POINTER(abi.Factor)]
        self._libc = C.CDLL(None)
        self._libc.calloc.restype = C.c_void_p
        self._libc.calloc.argtypes = [C.c_size_t, C.c_size_t]

    # -- product-only helpers -------------------------------------------------------------------
    def version(self):
        return self.dll.aprilsam_amd_version().decode()

    def device_count(self):
        return int(self.dll.aprilsam_amd_device_count())

    def set_option(self, name, value):
        rc = self.dll.aprilsam_amd_set_option(name.encode(), float(value))
        if rc != 0:
            raise ValueError(f"unknown option {name}")

    def get_option(self, name):
        v = C.c_double()
        if self.dll.aprilsam_amd_get_option(name.encode(), C.byref(v)) != 0:
            raise ValueError(f"unknown option {name}")
        return v.value

    def options(self, **kw):
        """context manager: set the given options, restore what they were on the way out"""
        lib = self

        class _Scope:
            def __enter__(self):
                self.saved = {k: lib.get_option(k) for k in kw}
                for k, v in kw.items():
                    lib.set_option(k, v)
                return lib

            def __exit__(self, *exc):
                for k, v in self.saved.items():
                    lib.set_option(k, v)
                return False
        return _Scope()

    def last_error(self):
        """(code, message) of the most recent failed call in this process; (0, "") when there was none"""
        buf = C.create_string_buffer(1024)
        code = int(self.dll.aprilsam_amd_last_error(buf, 1024))
        return code, buf.value.decode()

    def clear_error(self):
        self.dll.aprilsam_amd_clear_error()

    def lattice_arrays(self, K):
        """SURVEY.md §8(d) synthetic Manhattan lattice as arrays (states, fa, fb, z, W)."""
        N = K * K
        F = 2 * K * (K - 1) + 2 * (K - 1) * (K - 1) + 1
        states = np.zeros((N, 3)); fa = np.zeros(F, np.int32); fb = np.zeros(F, np.int32)
        z = np.zeros((F, 3)); W = np.zeros((F, 9))
        n = self.dll.aprilsam_amd_lattice_arrays(K, _np_d(states), _np_i(fa), _np_i(fb), _np_d(z), _np_d(W))
        assert n == F, (n, F)
        return states, fa, fb, z, W

    def new_graph(self):
        return Graph(self)

    def load_graph(self, path):
        """april_graph_create_from_file (april_graph.c:398-426); None when the file cannot be read"""
        if not self.is_product:
            self.dll.april_graph_stype_init(); self.dll.stype_register_basic_types()
        p = self.dll.april_graph_create_from_file(os.fsencode(path))
        if not p:
            return None
        g = Graph.__new__(Graph); g.lib = self; g.ptr = p
        return g

    def new_param(self, **kw):
        return Param(self, **kw)


class Param:
    """april_graph_cholesky_param_t, heap allocated as the reference demands (its _destory frees it)."""

    def __init__(self, lib, nthreshold=100, delta_xy=0.1, delta_theta=0.1, show_timing=0):
        self.lib = lib
        mem = lib._libc.calloc(1, C.sizeof(abi.CholeskyParam))
        self.ptr = C.cast(mem, C.POINTER(abi.CholeskyParam))
        lib.dll.april_graph_cholesky_param_init(self.ptr)
        p = self.ptr.contents
        p.nthreshold, p.delta_xy, p.delta_theta, p.show_timing = nthreshold, delta_xy, delta_theta, show_timing

    @property
    def c(self):
        return self.ptr.contents

    def stats(self):
        st = abi.Stats()
        rc = self.lib.dll.aprilsam_amd_get_stats(self.ptr, C.byref(st))
        if rc != 0:
            raise RuntimeError("no solver context for this param yet")
        return st.asdict()

    def destroy(self):
        if self.ptr:
            self.lib.dll.april_graph_cholesky_param_destory(self.ptr)
            self.ptr = None


class Graph:
    def __init__(self, lib):
        self.lib = lib
        self.ptr = lib.dll.april_graph_create()

    # -- construction (aprilsam.h:285-288) --------------------------------------------------------
    def add_node_xyt(self, state, init=None, truth=None):
        s = (C.c_double * 3)(*state)
        i = (C.c_double * 3)(*(init if init is not None else state))
        t = (C.c_double * 3)(*(truth if truth is not None else state))
        n = self.lib.dll.april_graph_node_xyt_create(s, i, t)
        self.lib._add_node(self.ptr, n)
        return self.n_nodes - 1

    @staticmethod
    def _matd(W):
        m = abi.Matd3x3()
        m.nrows = m.ncols = 3
        for k, v in enumerate(np.asarray(W, float).reshape(9)):
            m.data[k] = v
        return m

    def add_factor_xyt(self, a, b, z, W):
        zz = (C.c_double * 3)(*z)
        m = self._matd(W)
        f = self.lib.dll.april_graph_factor_xyt_create(int(a), int(b), zz, None, C.byref(m))
        self.lib._add_factor(self.ptr, f)

    def add_factor_xytpos(self, a, z, W):
        zz = (C.c_double * 3)(*z)
        m = self._matd(W)
        f = self.lib.dll.april_graph_factor_xytpos_create(int(a), zz, None, C.byref(m))
        self.lib._add_factor(self.ptr, f)

    def build_from_arrays(self, states, fa, fb, z, W):
        """Bulk append: N nodes, F factors (fb<0 => xytpos prior on fa)."""
        states = np.ascontiguousarray(states, float); z = np.ascontiguousarray(z, float)
        W = np.ascontiguousarray(W, float).reshape(-1, 9)
        fa = np.ascontiguousarray(fa, np.int32); fb = np.ascontiguousarray(fb, np.int32)
        fn = self.lib.dll.aprilsam_amd_graph_from_arrays if self.lib.is_product else self.lib.dll.rs_build_from_arrays
        fn.argtypes = [C.POINTER(abi.Graph), C.c_int, _dp, C.c_int, _ip, _ip, _dp, _dp]
        fn(self.ptr, len(states), _np_d(states), len(fa), _np_i(fa), _np_i(fb), _np_d(z), _np_d(W))

    def save(self, path, magic_offset=0):
        """april_graph_save (april_graph.c:377-396): True on success"""
        if self.lib.is_product:
            return self.lib.dll.aprilsam_amd_graph_save_ex(self.ptr, os.fsencode(path), int(magic_offset)) == 1
        self.lib.dll.april_graph_stype_init(); self.lib.dll.stype_register_basic_types()
        return self.lib.dll.april_graph_save(self.ptr, os.fsencode(path)) == 1

    def factor_attr_put(self, i, key, value):
        f = self.factor(i)
        slot = C.cast(C.byref(f, abi.Factor.attr.offset), C.POINTER(C.c_void_p))
        rc = self.lib.dll.aprilsam_amd_attr_put_string(slot, key.encode(), value.encode())
        if rc != 0:
            raise RuntimeError(f"attr_put rc={rc}")

    def factor_attr_get(self, i, key):
        v = self.lib.dll.aprilsam_amd_attr_get_string(self.factor(i).attr, key.encode())
        return v.decode() if v is not None else None

    def arrays(self):
        """(states, fa, fb, z, W) of an xyt / xytpos graph, fb = -1 for priors"""
        n, F = self.n_nodes, self.n_factors
        fa = np.zeros(F, np.int32); fb = np.full(F, -1, np.int32); z = np.zeros((F, 3)); W = np.zeros((F, 9))
        for i in range(F):
            f = self.factor(i)
            fa[i] = f.nodes[0]
            if f.nnodes == 2:
                fb[i] = f.nodes[1]
            zz = f.u.z; Wd = f.u.W.contents.data
            for k in range(3):
                z[i, k] = zz[k]
            for k in range(9):
                W[i, k] = Wd[k]
        return self.states(), fa, fb, z, W

    # -- accessors ------------------------------------------------------------------------------
    @property
    def n_nodes(self):
        return self.ptr.contents.nodes.contents.size

    @property
    def n_factors(self):
        return self.ptr.contents.factors.contents.size

    def node(self, i):
        arr = C.cast(self.ptr.contents.nodes.contents.data, C.POINTER(C.POINTER(abi.Node)))
        return arr[i].contents

    def factor(self, i):
        arr = C.cast(self.ptr.contents.factors.contents.data, C.POINTER(C.POINTER(abi.Factor)))
        return arr[i].contents

    def _gather(self, field):
        n = self.n_nodes
        out = np.empty((n, 3))
        if self.lib.is_product and hasattr(self.lib.dll, "aprilsam_amd_graph_node_arrays"):      # one call instead of n ctypes round trips
            k = ("state", "l_point", "delta_X").index(field)
            args = [None, None, None]; args[k] = _np_d(out)
            self.lib.dll.aprilsam_amd_graph_node_arrays(self.ptr, *args)
            return out
        arr = C.cast(self.ptr.contents.nodes.contents.data, C.POINTER(C.POINTER(abi.Node)))
        for i in range(n):
            p = getattr(arr[i].contents, field)
            out[i, 0], out[i, 1], out[i, 2] = p[0], p[1], p[2]
        return out

    def states(self):
        return self._gather("state")

    def l_points(self):
        return self._gather("l_point")

    def deltas(self):
        return self._gather("delta_X")

    def states_of(self, i):
        st = self.node(i).state
        return [st[0], st[1], st[2]]

    def set_state(self, i, xyt, relinearize=False):
        nd = self.node(i)
        for k in range(3):
            nd.state[k] = xyt[k]
            if relinearize:
                nd.l_point[k] = xyt[k]

    def set_all_states(self, states, relinearize=False):
        for i in range(self.n_nodes):
            self.set_state(i, states[i], relinearize)

    def set_all_W(self, W):
        """edit the information matrices of all xyt / xytpos factors in place (W: [F, 9])"""
        W = np.asarray(W, float).reshape(-1, 9)
        for i in range(self.n_factors):
            Wd = self.factor(i).u.W.contents.data
            for k in range(9):
                Wd[k] = W[i, k]

    # -- solver (aprilsam.h:268-281) -----------------------------------------------------------------
    def chi2(self):
        return float(self.lib.dll.april_graph_chi2(self.ptr))

    def cholesky(self, param):
        self.lib.dll.april_graph_cholesky(self.ptr, param.ptr)

    def cholesky_inc(self, param):
        self.lib.dll.april_graph_cholesky_inc(self.ptr, param.ptr)

    def cholesky_inc_solver(self, param):
        """april_graph_cholesky_inc_solver (aprilsam.c:578-597); neither library reads the idxs argument"""
        self.lib.dll.april_graph_cholesky_inc_solver(self.ptr, param.ptr, None)

    def batch_resident(self, param, iters):
        chi2 = np.zeros(iters + 1); ms = np.zeros(iters)
        rc = self.lib.dll.aprilsam_amd_batch_resident(self.ptr, param.ptr, iters, _np_d(chi2), _np_d(ms))
        if rc != 0:
            raise RuntimeError(f"aprilsam_amd_batch_resident failed rc={rc}")
        return chi2, ms

    def destroy(self):
        if self.ptr:
            self.lib.dll.april_graph_destroy(self.ptr)
            self.ptr = None
