#!/usr/bin/env python3
"""Aggregate the rocprofv3 csv files written by tools/profile_round.sh into the small files kept under profiles/:
    <tag>_kernel_stats.csv          per-kernel totals of the M3500 bench command
    <tag>_kernel_stats_lattice.csv  same for the 100k-lattice run
    <tag>_pmc_hbm.json              FETCH_SIZE / WRITE_SIZE per kernel (raw KB, per dispatch)
    <tag>_pmc_mfma.json             MFMA busy cycles / CU busy cycles / F64 MFMA ops per kernel (100k lattice)
usage: pmc_aggregate.py <dir written by profile_round.sh> <tag>"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
SRC_HASH = bench.source_hash()
dst = os.path.join(src, "summary"); os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    f = sorted(glob.glob(os.path.join(src, sub, "**", pat), recursive=True))
    return f[0] if f else None


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:48]


def counters(sub):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return None
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            a = acc[row["Counter_Name"]][short(row["Kernel_Name"])]
            a[0] += 1; a[1] += float(row["Counter_Value"])
    return {c: {k: {"dispatches": v[0], "total": round(v[1], 1), "per_dispatch": round(v[1] / v[0], 2)} for k, v in ks.items()} for c, ks in acc.items()}


for sub, out in (("stats", f"{tag}_kernel_stats.csv"), ("stats_lattice", f"{tag}_kernel_stats_lattice.csv"), ("stats_lattice1m", f"{tag}_kernel_stats_lattice1m.csv"),
                 ("stats_inc", f"{tag}_kernel_stats_inc_demo.csv")):
    f = find(sub, "*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, out)); print("wrote", out)
hbm = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    r = counters("pmc_" + c)
    if r and c in r:
        hbm[c] = {k: {"dispatches": v["dispatches"], "total_kb": v["total"], "kb_per_dispatch": v["per_dispatch"]} for k, v in r[c].items()}
if hbm:
    json.dump({"source_hash": SRC_HASH, "command": "APRILSAM_AMD_USE_GRAPH=0 rocprofv3 --pmc <COUNTER> --output-format csv -- python bench.py --steps 10 --warmup 2 --no-lattice "
                          "--no-inc --no-cpu-baseline (one pass per counter, hipGraph replay off)",
               "unit": "KB as reported by rocprofv3; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 - "
                       "values are RAW, bench.py applies the x2 to the read side", "counters": hbm},
              open(os.path.join(dst, f"{tag}_pmc_hbm.json"), "w"), indent=1)
    print("wrote", f"{tag}_pmc_hbm.json")
for suffix, label, out in (("lat316", "python tools/lattice_big.py 316 2 (100k-pose lattice, config 4)", f"{tag}_pmc_hbm_lattice100k.json"),
                           ("lat1000", "python tools/lattice_big.py 1000 1 (1M-pose lattice, config 5)", f"{tag}_pmc_hbm_lattice1m.json")):
    hb = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        r = counters(f"pmc_{c}_{suffix}")
        if r and c in r:
            hb[c] = {k: {"dispatches": v["dispatches"], "total_kb": v["total"], "kb_per_dispatch": v["per_dispatch"]} for k, v in r[c].items()}
    if hb:
        json.dump({"source_hash": SRC_HASH, "command": "APRILSAM_AMD_USE_GRAPH=0 rocprofv3 --pmc <COUNTER> --output-format csv -- " + label + ", one pass per counter",
                   "unit": "KB as reported by rocprofv3, RAW; bench.py applies the gfx950 x2 on the read side (MI355X_MICROARCH.md)", "counters": hb},
                  open(os.path.join(dst, out), "w"), indent=1)
        print("wrote", out)
try:
    b = json.loads(open(os.path.join(src, "bench_cpu_lattice100k.json")).read().strip().splitlines()[-1])
    rc = b["lattice100k"]["reference_cpu_same_host"]
    json.dump({k: rc[k] for k in ("s_per_iter", "cores", "chi2_after_1", "host")}, open(os.path.join(dst, f"{tag}_cpu_lattice100k.json"), "w"), indent=1)
    print("wrote", f"{tag}_cpu_lattice100k.json")
except Exception as e:
    print("no cpu lattice100k record:", repr(e))
for sub, label, out in (("pmc_mfma", "python tools/lattice_big.py 316 2   (100k-pose lattice, config 4", f"{tag}_pmc_mfma.json"),
                        ("pmc_mfma1m", "python tools/lattice_big.py 1000 1   (1M-pose lattice, config 5", f"{tag}_pmc_mfma_lattice1m.json")):
    m = counters(sub)
    if not m:
        continue
    per = {}
    for k in set().union(*[set(v) for v in m.values()]):
        g = lambda c: m.get(c, {}).get(k, {}).get("total", 0.0)
        busy, cu, ops = g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_BUSY_CU_CYCLES"), g("SQ_INSTS_VALU_MFMA_MOPS_F64")
        per[k] = {"dispatches": m[next(iter(m))].get(k, {}).get("dispatches", 0), "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CU_CYCLES": cu,
                  "SQ_INSTS_VALU_MFMA_MOPS_F64": ops, "GRBM_GUI_ACTIVE": g("GRBM_GUI_ACTIVE"),
                  "mfma_busy_over_cu_busy": round(busy / cu, 4) if cu else None,
                  "cus_busy_on_average": round(cu / g("GRBM_GUI_ACTIVE"), 2) if g("GRBM_GUI_ACTIVE") else None}
    json.dump({"source_hash": SRC_HASH,
               "command": "APRILSAM_AMD_USE_GRAPH=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE "
                          "-- " + label + "; sums over all dispatches of the run)",
               "kernels": per}, open(os.path.join(dst, out), "w"), indent=1)
    print("wrote", out)
for f in ("ubench_mfma_f64.txt", "ubench_valu_lat.txt", "ubench_launch_lat.txt", "chain_times_lattice100k.txt", "inc_profile.txt", "inc_trace_medians.txt", "inc_slowest.txt",
          "first_call.txt", "plan_time.txt", "inc_steps.txt", "inc_steps_no_update.txt", "batch_only.txt", "stats_lattice.log", "stats_lattice1m.log"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}")); print("wrote", f"{tag}_{f}")
