"""debug: RCCL transport with world = 1, with / without torch in the process: python tools/rccl_probe.py [torch|torchcuda]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode.startswith("torch"):
    import torch
    if mode == "torchcuda":
        torch.cuda.init(); torch.zeros(4, device="cuda")
from aprilsam_amd import host
lib = host.SolverLib()
g = lib.new_graph(); lib.dll.aprilsam_amd_make_lattice(g.ptr, 40); p = lib.new_param()
d = lib.dll
gp, pp = C.cast(g.ptr, C.c_void_p), C.cast(p.ptr, C.c_void_p)
d.aprilsam_amd_shard_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
d.aprilsam_amd_shard_iterate.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
d.aprilsam_amd_shard_gather_states.argtypes = [C.c_void_p, C.c_void_p]
d.aprilsam_amd_shard_chi2.restype = C.c_double; d.aprilsam_amd_shard_chi2.argtypes = [C.c_void_p, C.c_void_p]
d.aprilsam_amd_shard_comm_init_rccl.argtypes = [C.c_void_p, C.c_char_p]
print("begin", d.aprilsam_amd_shard_begin(gp, pp, 0, 1))
buf = C.create_string_buffer(128)
print("id", d.aprilsam_amd_shard_comm_unique_id(buf))
print("init", d.aprilsam_amd_shard_comm_init_rccl(pp, buf))
print("chi2", d.aprilsam_amd_shard_chi2(gp, pp))
print("iterate", d.aprilsam_amd_shard_iterate(gp, pp, 2))
print("chi2", d.aprilsam_amd_shard_chi2(gp, pp))
print("gather", d.aprilsam_amd_shard_gather_states(gp, pp))
print("mode", mode, "ok")
