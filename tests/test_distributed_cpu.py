"""N > 1 paths on CPU.
(1) bench.py's replica logic with world_size-2 gloo processes (host side: plan of the M3500 replica, barrier,
    max-over-ranks timing, aggregate throughput).  No data-path collective — M3500 does not shard.
(2) the nested-dissection subtree sharding of large graphs: ownership map and exchange lists (host logic of
    aprilsam_amd_shard_begin, reachable without a GPU through aprilsam_amd_shard_plan), and the exchange SCHEDULE of
    aprilsam_amd_shard_iterate (csrc/solver.hip.cpp; host-callback transport = blocking send / recv / broadcast in list
    order) driven by two gloo ranks with stand-in payloads: same order of sends / receives / broadcasts as the library,
    every slab arrives where the parent lives, no deadlock."""
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


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_shard_map_is_a_consistent_partition(lib, world):
    from tests.support.mf_emulator import PlanView
    st, fa, fb, z, W = lib.lattice_arrays(100)
    P = PlanView(lib, len(st), fa, fb, xy=st[:, :2], leaf_nodes=16, shard_worlds=(world,))
    xfer, bcast, owner = P.shard[world]
    nF = P.nF
    assert len(owner) == nF and owner.min() >= 0 and owner.max() < world
    assert set(owner.tolist()) == set(range(world))                          # every rank owns something
    # transfers = exactly the tree edges that cross ranks, child -> owner of the parent, levels ascending
    cross = [(t, int(owner[t]), int(owner[P.front_parent[t]])) for t in range(nF) if P.front_parent[t] >= 0 and owner[P.front_parent[t]] != owner[t]]
    assert sorted((int(r[1]), int(r[2]), int(r[3])) for r in xfer) == sorted(cross)
    assert all(int(r[0]) == int(P.front_level[int(r[1])]) and int(r[5]) > 0 for r in xfer)
    # a subtree below a front owned exclusively by one rank never leaves that rank: owners only change along "top" fronts
    top = set(int(r[1]) for r in bcast)
    for t in range(nF):
        par = int(P.front_parent[t])
        if par >= 0 and par not in top:
            assert owner[t] == owner[par]
    if world > 1:
        assert len(xfer) >= world - 1 and len(top) >= world - 1
        # balance of the parallel phase: the subtrees below the top fronts (the top fronts themselves run level after
        # level whoever owns them).  Work proxy = flops of the fronts, as in the mapping.
        ns = 3.0 * P.front_nsb; m = 3.0 * (P.front_nsb + P.front_nub)
        local = np.array([t not in top for t in range(nF)])
        w = np.bincount(owner[local], weights=(ns * m * m)[local], minlength=world)
        assert w.min() > 0 and w.max() < 2.1 * w.mean(), w     # (binary tree over 3 ranks: 50 / 25 / 25 at best)
    else:
        assert len(xfer) == 0 and len(bcast) == 0


def _shard_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from aprilsam_amd import host
    from tests.support.mf_emulator import PlanView
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = host.SolverLib()
    st, fa, fb, z, W = lib.lattice_arrays(40)
    P = PlanView(lib, len(st), fa, fb, xy=st[:, :2], leaf_nodes=16, shard_worlds=(world,))
    xfer, bcast, owner = P.shard[world]
    nlev = int(P.front_level.max()) + 1
    got, sent = [], []
    # the schedule of aprilsam_amd_shard_iterate (host-callback transport) with stand-in payloads (front id, packed count)
    for l in range(nlev):
        for lev, front, src, dst, off, cnt in xfer[xfer[:, 0] == l]:
            buf = torch.zeros(3, dtype=torch.float64)
            if rank == src:
                buf[:] = torch.tensor([float(front), float(cnt), 1.0]); dist.send(buf, dst=int(dst)); sent.append(int(front))
            elif rank == dst:
                dist.recv(buf, src=int(src)); got.append((int(buf[0]), int(buf[1])))
                assert int(buf[0]) == int(front) and int(buf[1]) == int(cnt)
    for l in range(nlev - 1, -1, -1):
        for lev, front, own, first, nsb in bcast[bcast[:, 0] == l]:
            buf = torch.full((3,), float(front) if rank == own else -1.0, dtype=torch.float64)
            dist.broadcast(buf, src=int(own))
            assert float(buf[0]) == float(front)
    out.put((rank, sorted(sent), sorted(g[0] for g in got), len(xfer), len(bcast)))
    dist.barrier(); dist.destroy_process_group()


def test_shard_exchange_schedule_two_ranks_gloo(built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29900 + os.getpid() % 2000
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sent0, got0, nx, nb), (_, sent1, got1, nx1, nb1) = res
    assert (nx, nb) == (nx1, nb1) and nx >= 1
    assert sorted(sent0 + sent1) == sorted(got0 + got1) and len(sent0) + len(sent1) == nx      # every slab sent once, received once
