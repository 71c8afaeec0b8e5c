"""Launcher of the multi-GPU path: one process per GPU.  The library owns everything that matters -- plan, ownership
map, per-rank front pool, kernels AND the exchange (include/aprilsam_amd.h, aprilsam_amd_shard_*): with backend "nccl"
it talks RCCL over xGMI itself, on its own HIP stream (ncclSend / ncclRecv of the packed Schur slabs, ncclBroadcast of
the separator solutions); this module only hands it the RCCL unique id.  With backend "gloo" (the tests: several ranks
sharing one GPU, which RCCL cannot do) the library stages every buffer through pinned host memory and calls back into
the four functions below, which move it with torch.distributed.

There is no all-reduce on the data path; chi^2 and the final gather of the states (outside the timed region) are sums
over the ranks.
"""
import ctypes as C

import numpy as np

_SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_longlong, C.c_int)
_RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_longlong, C.c_int)
_BCAST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_longlong, C.c_int)
_ALLRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_longlong)


class HostComm(C.Structure):
    """aprilsam_amd_host_comm_t"""
    _fields_ = [("user", C.c_void_p), ("send", _SEND), ("recv", _RECV), ("bcast", _BCAST), ("allreduce_sum", _ALLRED)]


class ShardedSolver:
    def __init__(self, lib, graph, param, rank, world, backend="nccl", device=None, _pretend_rccl_unavailable_on=None):
        # (_pretend_rccl_unavailable_on: tests only -- that rank reports "librccl not loadable", to walk the fall-back)
        self._pretend_unavailable = _pretend_rccl_unavailable_on
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.lib, self.g, self.p, self.rank, self.world = lib, graph, param, rank, world
        self.backend = backend
        d = lib.dll
        d.aprilsam_amd_shard_info.restype = C.c_longlong
        d.aprilsam_amd_shard_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.c_longlong]
        d.aprilsam_amd_shard_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        d.aprilsam_amd_shard_iterate.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        d.aprilsam_amd_shard_gather_states.argtypes = [C.c_void_p, C.c_void_p]
        d.aprilsam_amd_shard_chi2.restype = C.c_double
        d.aprilsam_amd_shard_chi2.argtypes = [C.c_void_p, C.c_void_p]
        d.aprilsam_amd_shard_comm_unique_id.argtypes = [C.c_char_p]
        d.aprilsam_amd_shard_comm_init_rccl.argtypes = [C.c_void_p, C.c_char_p]
        d.aprilsam_amd_shard_comm_init_host.argtypes = [C.c_void_p, C.POINTER(HostComm)]
        d.aprilsam_amd_shard_end.argtypes = [C.c_void_p]
        self._g = C.cast(graph.ptr, C.c_void_p); self._p = C.cast(param.ptr, C.c_void_p)
        rc = d.aprilsam_amd_shard_begin(self._g, self._p, rank, world)
        if rc != 0:
            raise RuntimeError(f"shard_begin failed rc={rc}")
        self.n_levels, self.n_fronts, self.n_nodes, self.pool_doubles, self.pool_doubles_all = (int(v) for v in self._info(0))
        self.xfer = self._info(1).reshape(-1, 6)       # level, front, src, dst, -, packed count
        self.bcast = self._info(2).reshape(-1, 5)      # level, front, owner, first position, blocks
        self.owner = self._info(3)
        self.transport_note = None
        if world > 1 or backend == "nccl":
            if backend == "nccl":
                # RCCL inside the library; if the communicator cannot be created on ANY rank (librccl not loadable next to this
                # HIP runtime, a clash with the RCCL the host framework bundles, ...) every rank falls back to the host-callback
                # transport over a gloo group -- slower (pinned staging), but the same schedule and the same results
                err = None
                try:
                    self._init_rccl(device)
                except Exception as e:                # noqa: BLE001
                    err = repr(e)
                ok = self._all_ok(err is None, device)
                if not ok:
                    self.transport_note = f"RCCL transport unavailable ({err or 'failed on another rank'}): host callbacks over gloo instead"
                    if world > 1:
                        self._gloo = dist.new_group(backend="gloo")
                        self.backend = "gloo (fallback)"
                        self._init_host(self._gloo)
            else:
                self._init_host()

    def _all_ok(self, mine, device):
        """logical AND of `mine` over the ranks (through the process group the launcher set up)"""
        if self.world <= 1:
            return mine
        torch, dist = self.torch, self.dist
        dev = self._coll_device(device)
        t = torch.tensor([1 if mine else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def _coll_device(self, device):
        """where tensors of the launcher's own collectives live: the (current) GPU under an nccl group, the host otherwise -- one
        rule for every collective of the set-up, so that no rank can take a different turn"""
        torch, dist = self.torch, self.dist
        if dist.get_backend() != "nccl":
            return "cpu"
        return device if device is not None else torch.device("cuda", torch.cuda.current_device())

    # -- transports ---------------------------------------------------------------------------------------------------
    def _init_rccl(self, device):
        """rank 0 asks the library for an RCCL unique id; torch.distributed only carries these 128 bytes.  Before the collective
        ncclCommInitRank every rank checks that IT can load RCCL at all (a rank that cannot would leave the others waiting
        inside the collective): the check is the id call itself, into a scratch buffer, and its outcome is AND-ed over the ranks"""
        torch, dist = self.torch, self.dist
        buf = C.create_string_buffer(128)
        rc = self.lib.dll.aprilsam_amd_shard_comm_unique_id(buf)          # (loads librccl next to the library's HIP runtime)
        if self._pretend_unavailable is not None and int(self._pretend_unavailable) == self.rank:
            rc = -5
        if not self._all_ok(rc == 0, device):
            raise RuntimeError(f"librccl not usable on every rank (shard_comm_unique_id rc={rc} here)")
        if self.world > 1:
            t = torch.tensor(list(buf.raw), dtype=torch.uint8, device=self._coll_device(device))
            dist.broadcast(t, src=0)                                        # rank 0's id wins
            buf = C.create_string_buffer(bytes(t.cpu().tolist()), 128)
        rc = self.lib.dll.aprilsam_amd_shard_comm_init_rccl(self._p, buf)
        if rc != 0:
            raise RuntimeError(f"shard_comm_init_rccl rc={rc}")

    def _init_host(self, group=None):
        torch, dist = self.torch, self.dist

        def view(ptr, n):
            return torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(int(n),)))

        def guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:        # never let an exception cross the C boundary
                    print("aprilsam_amd.shard: communication callback failed:", repr(e), flush=True)
                    return 1
            return run

        self._cb = HostComm(None,
                            _SEND(guard(lambda u, p, n, dst: dist.send(view(p, n), dst=int(dst), group=group))),
                            _RECV(guard(lambda u, p, n, src: dist.recv(view(p, n), src=int(src), group=group))),
                            _BCAST(guard(lambda u, p, n, root: dist.broadcast(view(p, n), src=int(root), group=group))),
                            _ALLRED(guard(lambda u, p, n: dist.all_reduce(view(p, n), group=group))))
        rc = self.lib.dll.aprilsam_amd_shard_comm_init_host(self._p, C.byref(self._cb))
        if rc != 0:
            raise RuntimeError(f"shard_comm_init_host rc={rc}")

    def _info(self, what):
        n = self.lib.dll.aprilsam_amd_shard_info(self._p, what, None, 0)
        out = np.zeros(max(int(n), 1), np.int64)
        self.lib.dll.aprilsam_amd_shard_info(self._p, what, out.ctypes.data_as(C.POINTER(C.c_longlong)), n)
        return out[:n]

    # -- the solve ----------------------------------------------------------------------------------------------------
    def iterate(self, n=1):
        rc = self.lib.dll.aprilsam_amd_shard_iterate(self._g, self._p, int(n))
        if rc == -2:
            raise ArithmeticError("sharded solve: not positive definite")
        if rc != 0:
            raise RuntimeError(f"shard_iterate rc={rc}")

    def chi2(self):
        return float(self.lib.dll.aprilsam_amd_shard_chi2(self._g, self._p))

    def gather_states(self):
        """every rank ends up with all states (device arrays and node objects); returns them as an (N, 3) array"""
        rc = self.lib.dll.aprilsam_amd_shard_gather_states(self._g, self._p)
        if rc != 0:
            raise RuntimeError(f"shard_gather_states rc={rc}")
        return self.g.states()

    def comm_info(self):
        """what the attached transport is, as RCCL itself reports it (aprilsam_amd_shard_comm_info)"""
        out = (C.c_longlong * 5)(); path = C.create_string_buffer(512)
        self.lib.dll.aprilsam_amd_shard_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_char_p, C.c_int]
        self.lib.dll.aprilsam_amd_shard_comm_info(self._p, out, path, 512)
        kind = {0: "none (single rank)", 1: "RCCL point-to-point on the solver stream", 2: "host callbacks over " + self.backend}[int(out[0])]
        v = int(out[3])
        if self.transport_note:
            kind += " -- " + self.transport_note
        return {"transport": kind, "ncclCommCount": int(out[1]), "ncclCommUserRank": int(out[2]),
                "rccl_version": f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v else None, "librccl": path.value.decode() or None,
                "hip_device": int(out[4])}

    def modelled_speedup(self):
        """bound on the speed-up of the factorisation under this mapping: fronts that span several ranks are serial (one owner each),
        below them every rank works through its own subtrees (aprilsam_amd_shard_info what = 4)"""
        total, path, local, topall = (float(v) for v in self._info(4))
        return {"flops_total": total, "flops_heaviest_root_path_of_top_fronts": path, "flops_busiest_rank_subtrees": local,
                "flops_all_top_fronts": topall, "speedup_bound": total / max(path + local, 1.0)}

    def comm_bytes_per_iteration(self):
        return int(8 * (self.xfer[:, 5].sum() + 3 * self.bcast[:, 4].sum() * (self.world - 1)))

    def close(self):
        self.lib.dll.aprilsam_amd_shard_end(self._p)
