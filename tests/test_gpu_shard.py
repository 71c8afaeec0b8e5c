"""Multi-GPU path, functional check on ONE GPU: two ranks (gloo, host staging) share cuda:0 and solve a lattice by
nested-dissection subtree sharding; the result must equal the reference golden (and therefore the single-GPU
path).  On the 8-GPU node the same driver runs with backend "nccl" (RCCL)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, K, iters, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from aprilsam_amd import host
    from aprilsam_amd.shard import ShardedSolver
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = host.SolverLib()
    arr = lib.lattice_arrays(K)
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    sol = ShardedSolver(lib, g, p, rank, world, backend="gloo")
    chi2 = [sol.chi2()]
    for _ in range(iters):
        sol.iterate(1)
        chi2.append(sol.chi2())
    owned_fronts = int((sol.owner == rank).sum())
    out.put((rank, chi2, owned_fronts, int(sol.n_fronts), len(sol.xfer), sol.comm_bytes_per_iteration()))
    sol.close(); p.destroy(); g.destroy()
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_lattice_two_ranks_one_gpu(built, world):
    import torch.multiprocessing as mp
    K = 24
    G = np.load(os.path.join(ROOT, "tests", "golden", f"lattice_{K}.npz"))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29600 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, len(G["chi2"]) - 1, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, chi2, owned, nf, nx, comm in res:
        assert np.max(np.abs(np.array(chi2) - G["chi2"]) / G["chi2"]) < 1e-6, (rank, chi2)
        assert 0 < owned < nf and nx >= world - 1
    assert sum(r[2] for r in res) == res[0][3]               # every front has exactly one owner
