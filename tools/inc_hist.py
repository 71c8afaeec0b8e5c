#!/usr/bin/env python3
"""SURVEY.md section 8 row f4 -- "recent poses last": what a new factor dirties in the incremental path, with and without
keeping the newest poses in the root front (option pin_last, the counterpart of the reference's constrained ordering,
aprilsam.c:1021-1098).  Runs the 3500-step M3500 demo (deterministic schedule) for several pin_last values and writes
profiles/<tag>_inc_hist.json: per setting the distribution of regenerated fronts per step, the step-time distribution,
the same split by step class, and the cost of the steps right after a batch fall-back.

    python tools/inc_hist.py [tag]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aprilsam_amd import datasets, harness, host

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
lib = host.SolverLib()
states, fa, fb, z, W = datasets.m3500_arrays()
N = len(states)
G = np.load(os.path.join(ROOT, "tests", "golden", "m3500_inc_demo.npz"))
by_pose = [[] for _ in range(N)]
for k in range(len(fa)):
    by_pose[max(int(fa[k]), int(fb[k]))].append(k)


def run(pin):
    lib.set_option("pin_last", pin)
    g = lib.new_graph(); p = lib.new_param(nthreshold=100, delta_xy=0.1, delta_theta=0.1)
    ms = np.zeros(N); dirty = np.zeros(N, int); was_batch = np.zeros(N, bool); chi2 = np.zeros(N)
    since_batch = np.zeros(N, int); reach = np.zeros(N, int); last_batch = 0
    for k in range(N):
        g.add_node_xyt(states[k])
        if k == 0:
            g.add_factor_xytpos(0, [0, 0, 0], datasets.PRIOR_W)
        for f in (by_pose[k] if k else []):
            a, b = int(fa[f]), int(fb[f])
            if abs(a - b) == 1:
                if a < b:
                    g.set_state(b, harness._xyt_mul(g.states_of(a), z[f]), relinearize=True)
                else:
                    g.set_state(a, harness._xyt_mul(g.states_of(b), harness._xyt_inv(z[f])), relinearize=True)
            g.add_factor_xyt(a, b, z[f], W[f])
            reach[k] = max(reach[k], k - min(a, b))
        t0 = time.perf_counter()
        if k == 0:
            g.cholesky(p); was_batch[k] = True
        else:
            p.c.batch_time = 1e300
            bt = p.c.batch_time
            g.cholesky_inc(p)
            was_batch[k] = p.c.batch_time != bt
        ms[k] = (time.perf_counter() - t0) * 1e3
        st = p.stats()
        dirty[k] = st["reserved0"] if (not was_batch[k] and st["symbolic_reused"]) else -1
        if was_batch[k]:
            last_batch = k
        since_batch[k] = k - last_batch
        chi2[k] = g.chi2()
    p.destroy(); g.destroy()
    inc = ~was_batch
    fast = inc & (dirty >= 0)
    into_base = np.array([reach[k] > since_batch[k] for k in range(N)])       # a new factor reaches a pose of the frozen base plan

    def dist(x):
        x = np.asarray(x, float)
        return {"n": int(len(x)), "mean": float(x.mean()) if len(x) else None,
                "percentiles_10_25_50_75_90_99": [float(v) for v in np.percentile(x, [10, 25, 50, 75, 90, 99])] if len(x) else None}
    rel = np.abs(chi2 - G["chi2"]) / np.maximum(G["chi2"], 1e-9)
    return {
        "pin_last": pin, "total_ms": float(ms.sum()), "incremental_total_ms": float(ms[inc].sum()), "batch_fallbacks": int(was_batch.sum()) - 1,
        "fallback_schedule_identical_to_reference": bool(np.array_equal(was_batch, G["was_batch"])), "chi2_max_relerr": float(rel.max()),
        "step_ms": dist(ms[inc]), "regenerated_fronts_per_step": dist(dirty[fast]),
        "regenerated_fronts_histogram": {str(int(v)): int(c) for v, c in zip(*np.unique(dirty[fast], return_counts=True))},
        "steps_whose_new_factors_stay_in_the_tail": {"ms": dist(ms[fast & ~into_base]), "regenerated_fronts": dist(dirty[fast & ~into_base])},
        "steps_with_a_factor_into_the_base_plan": {"ms": dist(ms[fast & into_base]), "regenerated_fronts": dist(dirty[fast & into_base])},
        "first_step_after_a_batch": {"ms": dist(ms[fast & (since_batch == 1)]), "regenerated_fronts": dist(dirty[fast & (since_batch == 1)])},
        "steps_that_re_planned_instead": int((inc & (dirty < 0)).sum()),
    }


out = {"workload": "M3500 demo, 3500 poses one by one through april_graph_cholesky_inc (deterministic schedule), tail fronts of 24 poses",
       "graph_facts": {"loop_closures": int(sum(1 for k in range(len(fa)) if abs(int(fa[k]) - int(fb[k])) != 1)),
                       "poses_with_a_loop_closure": int(sum(1 for k in range(N) if any(abs(int(fa[f]) - int(fb[f])) != 1 for f in by_pose[k])))},
       "runs": [run(pin) for pin in (0, 8, 24, 64)]}
lib.set_option("pin_last", 0)
path = os.path.join(ROOT, "gpurun_out", f"{tag}_inc_hist.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
for r in out["runs"]:
    print(f"pin_last {r['pin_last']:3d}: total {r['total_ms']:.0f} ms, inc median {r['step_ms']['percentiles_10_25_50_75_90_99'][2]:.3f} ms, "
          f"regenerated fronts/step mean {r['regenerated_fronts_per_step']['mean']:.2f}, first step after batch: "
          f"{r['first_step_after_a_batch']['regenerated_fronts']['mean']:.2f} fronts / {r['first_step_after_a_batch']['ms']['mean']:.3f} ms, "
          f"schedule identical {r['fallback_schedule_identical_to_reference']}, re-planned {r['steps_that_re_planned_instead']}")
