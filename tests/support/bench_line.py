"""Shape of bench.py's ONE stdout line (TEST INFRASTRUCTURE): what the driver's parser must be able to take.  Round 5's line was 25.7 KB
with nested blocks and came back as `parsed: null`; the line is now contract fields + flat `config` / `roofline` / `cpu_baseline`, and
everything nested goes to bench_extras.json."""
import json

MAX_LINE_BYTES = 8192
MAX_CONFIG_KEYS = 50
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def check_line(s, cpu_baseline=True):
    assert "\n" not in s.strip()
    assert len(s.encode()) < MAX_LINE_BYTES, len(s.encode())
    assert "NaN" not in s and "Infinity" not in s
    out = json.loads(s)
    for k in CONTRACT:
        assert k in out, k
    scalar = (bool, int, float, str, type(None))
    for blk in ("config", "roofline") + (("cpu_baseline",) if cpu_baseline else ()):
        assert isinstance(out[blk], dict), blk
        for k, v in out[blk].items():
            assert isinstance(v, scalar), (blk, k, type(v))
    assert len(out["config"]) <= MAX_CONFIG_KEYS, len(out["config"])
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["frac"] > 0
    if cpu_baseline:
        c = out["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["value"] > 0 and c["kind"] in ("reference", "port")
    for k, v in out.items():                      # nothing nested outside the three blocks either
        assert isinstance(v, scalar) or k in ("config", "roofline", "cpu_baseline"), k
    return out
