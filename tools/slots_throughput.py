#!/usr/bin/env python3
"""Aggregate M3500 batch throughput of S device slots driven from S threads of ONE process (aprilsam_amd_param_set_device).
On a one-GPU box the slots share device 0 -- the numbers then say how much of the device one solve leaves idle (the M3500
iteration is a latency chain over ~30 workgroups, not a throughput problem); on a node with N devices slot s runs on device s.
    python tools/slots_throughput.py [slots ...]          (default 1 2 4 8)"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, host

lib = host.SolverLib()
arr = datasets.m3500_batch()
STEPS, CALLS = 400, 200


def worker(slot, barrier, out):
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    assert lib.dll.aprilsam_amd_param_set_device(p.ptr, slot) == 0
    d = lib.dll
    assert d.aprilsam_amd_resident_begin(g.ptr, p.ptr) == 0
    d.aprilsam_amd_resident_steps(g.ptr, p.ptr, 20, 0); d.aprilsam_amd_resident_sync(g.ptr, p.ptr)
    barrier.wait(); t0 = time.perf_counter()
    d.aprilsam_amd_resident_steps(g.ptr, p.ptr, STEPS, 0); rc = d.aprilsam_amd_resident_sync(g.ptr, p.ptr)
    t_res = time.perf_counter() - t0
    d.aprilsam_amd_resident_end(g.ptr, p.ptr); p.destroy(); g.destroy()
    g = lib.new_graph(); g.build_from_arrays(*arr); p = lib.new_param()
    assert lib.dll.aprilsam_amd_param_set_device(p.ptr, slot) == 0
    for _ in range(10):
        g.cholesky(p)
    barrier.wait(); t0 = time.perf_counter()
    for _ in range(CALLS):
        g.cholesky(p)
    t_api = time.perf_counter() - t0
    out[slot] = (rc, t_res, t_api, g.chi2(), lib.dll.aprilsam_amd_param_get_device(p.ptr))
    p.destroy(); g.destroy()


print(f"devices: {lib.device_count()}")
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    out = {}; barrier = threading.Barrier(S)
    th = [threading.Thread(target=worker, args=(s, barrier, out)) for s in range(S)]
    [t.start() for t in th]; [t.join(600) for t in th]
    assert len(out) == S and all(o[0] == 0 for o in out.values()), out
    t_res = max(o[1] for o in out.values()); t_api = max(o[2] for o in out.values())
    chi2 = {round(o[3], 6) for o in out.values()}
    print(f"slots {S}: resident {S * STEPS / t_res:9.0f} iterations/s aggregate ({1e3 * t_res / STEPS:.4f} ms per round of {S}), "
          f"API calls {S * CALLS / t_api:9.0f} /s aggregate ({1e3 * t_api / CALLS:.4f} ms per round), devices {sorted({o[4] for o in out.values()})}, chi2 {sorted(chi2)}",
          flush=True)
