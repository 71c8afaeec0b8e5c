"""ctypes mirror of the boundary structs declared in include/aprilsam_amd.h (PART 1).

Layout restated from the reference headers (aprilsam/aprilsam.h:65-72,98-146,151-179,231-265;
common/zarray.h:44-51; common/matd.h:46-51) — sizes/offsets are asserted in tests/test_abi.py
against SURVEY.md §8(b)'s measured values, and against objects created by the real reference
library when oracle/_ref is present.
"""
import ctypes as C


class ZArray(C.Structure):
    _fields_ = [("el_sz", C.c_size_t), ("size", C.c_int), ("alloc", C.c_int), ("data", C.c_void_p)]


class Matd3x3(C.Structure):
    """matd_t with a 3x3 payload (flexible array member materialised)."""
    _fields_ = [("nrows", C.c_uint), ("ncols", C.c_uint), ("data", C.c_double * 9)]


class Graph(C.Structure):
    _fields_ = [("factors", C.POINTER(ZArray)), ("nodes", C.POINTER(ZArray)),
                ("attr", C.c_void_p), ("stype", C.c_void_p)]


class FactorCommon(C.Structure):
    _fields_ = [("z", C.POINTER(C.c_double)), ("ztruth", C.POINTER(C.c_double)),
                ("W", C.POINTER(Matd3x3)), ("impl", C.c_void_p)]


class Factor(C.Structure):
    _fields_ = [("type", C.c_int), ("nnodes", C.c_int), ("nodes", C.POINTER(C.c_int)),
                ("length", C.c_int), ("attr", C.c_void_p),
                ("copy", C.c_void_p), ("eval", C.c_void_p), ("state_eval", C.c_void_p),
                ("destroy", C.c_void_p), ("u", FactorCommon), ("stype", C.c_void_p)]


class Node(C.Structure):
    _fields_ = [("UID", C.c_int), ("type", C.c_int), ("length", C.c_int),
                ("state", C.POINTER(C.c_double)), ("init", C.POINTER(C.c_double)),
                ("truth", C.POINTER(C.c_double)), ("l_point", C.POINTER(C.c_double)),
                ("delta_X", C.POINTER(C.c_double)), ("attr", C.c_void_p),
                ("copy", C.c_void_p), ("update", C.c_void_p), ("relinearize", C.c_void_p),
                ("destroy", C.c_void_p), ("impl", C.c_void_p), ("stype", C.c_void_p)]


class CholeskyParam(C.Structure):
    _fields_ = [("tikhanov", C.c_double), ("chol", C.c_void_p), ("factor_num", C.c_int),
                ("ordering", C.POINTER(C.c_int)), ("nreordering", C.c_int), ("show_timing", C.c_int),
                ("delta_x", C.POINTER(C.c_double)), ("B", C.POINTER(C.c_double)),
                ("y", C.POINTER(C.c_double)), ("A", C.c_void_p), ("tr", C.c_void_p),
                ("l_thresh", C.c_double), ("delta_thresh", C.c_double), ("nthreshold", C.c_int),
                ("batch_time", C.c_double), ("delta_xy", C.c_double), ("delta_theta", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("n_factors", C.c_int), ("n_fronts", C.c_int), ("n_levels", C.c_int),
                ("max_front_rows", C.c_int), ("symbolic_reused", C.c_int), ("not_spd", C.c_int),
                ("reserved0", C.c_int), ("nnz_L", C.c_longlong), ("flops_factor", C.c_double),
                ("bytes_fronts", C.c_double),
                ("ms_pack", C.c_double), ("ms_symbolic", C.c_double), ("ms_h2d", C.c_double),
                ("ms_device", C.c_double), ("ms_d2h", C.c_double), ("ms_unpack", C.c_double),
                ("ms_total", C.c_double), ("ms_dev_linearize", C.c_double), ("ms_dev_factor", C.c_double),
                ("ms_dev_solve", C.c_double), ("chi2_before", C.c_double), ("error_code", C.c_int), ("reserved1", C.c_int),
                ("inc_replanned", C.c_int), ("inc_old_old_cross", C.c_int), ("inc_fronts_updated", C.c_int), ("reserved2", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# measured LP64 layout of the reference; checked by tests/test_abi.py
EXPECTED_SIZES = {"ZArray": 24, "Graph": 32, "Factor": 104, "Node": 112, "CholeskyParam": 128}
EXPECTED_OFFSETS = {
    "Node": dict(UID=0, type=4, length=8, state=16, init=24, truth=32, l_point=40, delta_X=48, attr=56,
                 copy=64, update=72, relinearize=80, destroy=88, impl=96, stype=104),
    "Factor": dict(type=0, nnodes=4, nodes=8, length=16, attr=24, copy=32, eval=40, state_eval=48,
                   destroy=56, u=64, stype=96),
    "CholeskyParam": dict(tikhanov=0, chol=8, factor_num=16, ordering=24, nreordering=32, show_timing=36,
                          delta_x=40, B=48, y=56, A=64, tr=72, l_thresh=80, delta_thresh=88,
                          nthreshold=96, batch_time=104, delta_xy=112, delta_theta=120),
}
