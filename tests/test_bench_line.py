"""bench.py's stdout line from full records (CPU): size, shape, and that nothing nested survives in it.  The records are the committed
full outputs of earlier runs (profiles/*_bench.json: round 5's 25.7 KB line; profiles/*_bench_extras.json: the side file since round 6)."""
import glob
import json
import os

import pytest

import bench
from tests.support.bench_line import check_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]_bench.json")) + glob.glob(os.path.join(ROOT, "profiles", "*_bench_extras*.json")))


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_short_line_of_a_full_record(path):
    full = json.load(open(path))
    if "lattice100k" not in full and "api" not in full:
        pytest.skip("a short line, not a full record")
    s = json.dumps(bench.short_line(full))
    out = check_line(s, cpu_baseline="cpu_baseline" in full)
    cfg = out["config"]
    for k in ("api_call_it_per_s", "ms_per_api_call", "lattice100k_ms_per_step", "lattice1m_ms_per_step", "inc_total_ms", "batch_only_speedup"):
        assert k in cfg, k
    assert not any("_level" in k for k in cfg)                # no per-level keys ("levels" = the tree depth is one)


def test_short_line_survives_failed_extras_and_long_error_strings():
    full = json.load(open(RECORDS[0]))
    full["lattice1m"] = {"error": "watchdog: " + "x" * 300, "n_gpus": 8}
    full["lattice100k"] = {"error": "RuntimeError('" + "y" * 300 + "')"}
    full["m3500_incremental"] = {"error": "z" * 100}
    out = check_line(json.dumps(bench.short_line(full)), cpu_baseline="cpu_baseline" in full)
    assert out["config"]["lattice1m_error"].startswith("watchdog") and out["config"]["lattice1m_n_gpus"] == 8
