import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aprilsam_amd import host, harness
from tests.support.oracle_binding import REFLIB
from tests.support import asym_scenarios
lib = host.SolverLib(); ref = host.SolverLib(REFLIB)
arr = asym_scenarios.growth_graph()
def sym(a):
    st, fa, fb, z, W = a
    Ws = W.reshape(-1, 3, 3); Ws = ((Ws + Ws.transpose(0, 2, 1)) / 2).reshape(-1, 9)
    return st, fa, fb, z, Ws
for name, a, be in (("asym", arr, 250), ("asym-nobatch", arr, 0)):
    stats = {}
    def on_step(k, p, wb):
        s = p.stats(); stats[k] = (s["error_code"], s["not_spd"], s.get("inc_replanned", 0), s.get("reserved0", 0))
    ours = harness.run_demo(lib, a, deterministic=True, batch_every=be, on_step=on_step)
    theirs = harness.run_demo(ref, a, deterministic=True, batch_every=be)
    d = np.nonzero(ours["was_batch"] != theirs["was_batch"])[0]
    rel = np.abs(ours["chi2"] - theirs["chi2"]) / np.maximum(theirs["chi2"], 1e-9)
    bad = np.nonzero(rel > 1e-6)[0]
    print(name, "was_batch differs at", d[:20], "ours", ours["was_batch"][d[:20]], "| chi2 rel > 1e-6 at", bad[:12], rel[bad[:12]], "max", rel.max(), flush=True)
    print("   stats at first bad steps:", [(int(k), stats.get(int(k))) for k in list(d[:6]) + list(bad[:6])])
    print("   errors:", [(k, v) for k, v in stats.items() if v[0] or v[1]][:10], lib.last_error())
