"""N > 1 path of bench.py on CPU: world_size-2 gloo processes run the same replica logic (host side: plan of the
M3500 replica, barrier, max-over-ranks timing, aggregate throughput).  No GPU, no data-path collective — M3500
does not shard (replicas only)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from aprilsam_amd import datasets, host
    from tests.support.mf_emulator import PlanView
    barrier, max_over = bench.setup_dist(world, "gloo", torch.device("cpu"))
    lib = host.SolverLib()
    st, fa, fb, z, W = datasets.m3500_batch()
    P = PlanView(lib, len(st), fa, fb, xy=st[:, :2], leaf_nodes=16)       # every rank plans its own replica
    barrier()
    fake_dt = 0.5 + 0.25 * rank                                           # rank 1 is the slow one
    dt = max_over(fake_dt)
    out.put((rank, int(P.stats[0]), int(P.stats[3]), [int(v) for v in P.perm[:64]], dt, bench.aggregate_value(world, 100, dt)))
    barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_replicas_gloo(built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, nf0, nnz0, perm0, dt0, v0), (r1, nf1, nnz1, perm1, dt1, v1) = res
    assert (r0, r1) == (0, 1)
    assert (nf0, nnz0, perm0) == (nf1, nnz1, perm1)          # replicas are identical (deterministic planning)
    assert dt0 == dt1 == pytest.approx(0.75)                  # max over ranks
    assert v0 == v1 == pytest.approx(2 * 100 / 0.75)          # whole-job iterations / s
