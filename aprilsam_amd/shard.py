"""Host driver of the multi-GPU path: one process per GPU, torch.distributed for the exchange (backend "nccl" =
RCCL over xGMI on the 8-GPU node; "gloo" with host staging in the tests, where the ranks share one GPU).

The library (include/aprilsam_amd.h, aprilsam_amd_shard_*) owns the plan, the ownership map and every kernel;
this module only sequences the per-level steps and moves the two kinds of data that cross ranks:

  up   — the Schur update of a front whose parent lives on another rank, packed to the lower trapezoid the
         parent's assembly reads (k_pack_update): send/recv, point to point
  down — the solved x of the top fronts (a few thousand doubles each): broadcast

There is no all-reduce on the data path; chi^2 (outside the timed region) is one scalar all-reduce.
"""
import ctypes as C

import numpy as np


class ShardedSolver:
    def __init__(self, lib, graph, param, rank, world, backend="nccl", device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.lib, self.g, self.p, self.rank, self.world = lib, graph, param, rank, world
        self.on_gpu = backend == "nccl"
        self.dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if self.on_gpu else torch.device("cpu"))
        d = lib.dll
        d.aprilsam_amd_shard_info.restype = C.c_longlong
        d.aprilsam_amd_shard_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.c_longlong]
        d.aprilsam_amd_shard_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_int]
        d.aprilsam_amd_shard_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        d.aprilsam_amd_shard_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        d.aprilsam_amd_shard_chi2_local.restype = C.c_double
        d.aprilsam_amd_shard_chi2_local.argtypes = [C.c_void_p, C.c_void_p]
        d.aprilsam_amd_shard_end.argtypes = [C.c_void_p]
        self._g = C.cast(graph.ptr, C.c_void_p); self._p = C.cast(param.ptr, C.c_void_p)
        rc = d.aprilsam_amd_shard_begin(self._g, self._p, rank, world)
        if rc != 0:
            raise RuntimeError(f"shard_begin failed rc={rc}")
        self.n_levels, self.n_fronts, self.n_nodes = self._info(0)
        self.xfer = self._info(1).reshape(-1, 6)       # level, front, src, dst, pool offset, count
        self.bcast = self._info(2).reshape(-1, 5)      # level, front, owner, first position, blocks
        self.owner = self._info(3)
        self.up = [self.xfer[self.xfer[:, 0] == l] for l in range(self.n_levels)]
        self.down = [self.bcast[self.bcast[:, 0] == l] for l in range(self.n_levels)]
        nmax = int(max([1] + [int(r[5]) for r in self.xfer] + [3 * int(r[4]) for r in self.bcast]))
        # dbuf: device buffer the library packs into / unpacks from; buf: what torch.distributed moves (the same
        # tensor with RCCL, a host mirror with gloo)
        self.dbuf = torch.empty(nmax, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        self.buf = self.dbuf if self.on_gpu else torch.empty(nmax, dtype=torch.float64, device="cpu")

    def _info(self, what):
        n = self.lib.dll.aprilsam_amd_shard_info(self._p, what, None, 0)
        out = np.zeros(max(int(n), 1), np.int64)
        self.lib.dll.aprilsam_amd_shard_info(self._p, what, out.ctypes.data_as(C.POINTER(C.c_longlong)), n)
        return out[:n]

    def _step(self, op, arg=0):
        rc = self.lib.dll.aprilsam_amd_shard_step(self._g, self._p, op, arg)
        if rc != 0:
            raise RuntimeError(f"shard_step({op},{arg}) rc={rc}")

    def _copy(self, kind, off, cnt, to_lib):
        """kind 2: packed Schur update of front `off`; kind 1: x at elimination offset `off`.  Synchronous."""
        cnt = int(cnt)
        if to_lib and not self.on_gpu:
            self.dbuf[:cnt].copy_(self.buf[:cnt]); self.torch.cuda.synchronize()
        rc = self.lib.dll.aprilsam_amd_shard_copy(self._g, self._p, kind, int(off), cnt, C.c_void_p(self.dbuf.data_ptr()), 1 if to_lib else 0)
        if rc != 0:
            raise RuntimeError(f"shard_copy(kind={kind}) rc={rc}")
        if not to_lib and not self.on_gpu:
            self.buf[:cnt].copy_(self.dbuf[:cnt])

    def _comm_done(self):
        """With RCCL a collective / send / recv returns once it is ENQUEUED (the wait only chains the current torch stream
        behind the communication stream).  The library packs and unpacks on its own HIP stream, which torch knows nothing
        about, so the host has to wait for the communication before the buffer is read or overwritten."""
        if self.on_gpu:
            self.torch.cuda.current_stream().synchronize()

    def iterate(self, n=1):
        dist = self.dist
        for _ in range(n):
            self._step(0)
            for l in range(self.n_levels):
                self._step(1, l)
                for _, front, src, dst, off, cnt in self.up[l]:
                    if self.rank == src:
                        self._copy(2, front, cnt, False)
                        dist.send(self.buf[:cnt], dst=int(dst)); self._comm_done()
                    elif self.rank == dst:
                        dist.recv(self.buf[:cnt], src=int(src)); self._comm_done()
                        self._copy(2, front, cnt, True)
            for l in range(self.n_levels - 1, -1, -1):
                self._step(2, l)
                for _, front, owner, first, nsb in self.down[l]:
                    cnt = 3 * int(nsb)
                    if self.rank == owner:
                        self._copy(1, 3 * int(first), cnt, False)
                    dist.broadcast(self.buf[:cnt], src=int(owner)); self._comm_done()
                    if self.rank != owner:
                        self._copy(1, 3 * int(first), cnt, True)
            self._step(3)
        rc = self.lib.dll.aprilsam_amd_shard_step(self._g, self._p, 4, 0)
        if rc != 0:
            raise ArithmeticError("sharded solve: not positive definite")

    def chi2(self):
        t = self.torch.tensor([self.lib.dll.aprilsam_amd_shard_chi2_local(self._g, self._p)], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t)
        return float(t.item())          # .item() synchronises

    def comm_bytes_per_iteration(self):
        return int(8 * (self.xfer[:, 5].sum() + 3 * self.bcast[:, 4].sum() * (self.world - 1)))

    def close(self):
        self.lib.dll.aprilsam_amd_shard_end(self._p)
