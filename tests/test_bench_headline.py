"""bench.py's stdout contract: the LAST line is one compact JSON object (< 4000 bytes) that carries the bench keys, `roofline`
and `cpu_baseline` — round 2's 22 KB line did not survive the driver's stdout tail (BENCH_r02.parsed = null)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _canned():
    # a complete record of the default run as round 2 printed it (every leg, per-kernel tables, queue counts)
    with open(os.path.join(ROOT, "profiles", "r02", "bench_default.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_headline_is_compact_and_complete():
    b = _bench()
    full = _canned()
    assert len(json.dumps(full)) > 15000          # the blob really is the oversized record
    line = b.headline(full, "gpurun_out/bench_full.json")
    assert "\n" not in line and len(line) < 4000, len(line)
    h = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in h, k
    assert h["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "launches_per_step", "bytes_per_launch"):
        assert k in h["roofline"], k
    assert abs(h["roofline"]["frac"] - h["roofline"]["achieved"] / h["roofline"]["peak"]) < 1e-4
    for k in ("frac", "active_lane_frac"):
        assert k in h["roofline"]["issue"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in h["cpu_baseline"], k
    assert h["closest_hit"]["unit"] == "Mrays/s" and "frac" in h["closest_hit"]["roofline"]
    assert abs(h["value"] - full["value"]) <= 1e-6 * full["value"]


def test_headline_sheds_optional_legs_before_it_outgrows_the_tail():
    b = _bench()
    full = _canned()
    for key in ("c4_path", "c5_path", "closest_hit"):
        full[key]["workload"] = "x" * 5000
        full[key]["cpu_baseline"]["sample"] = "y" * 5000
        full[key]["roofline"]["kernel"] = "k" * 800
    full["config"]["workload"] = "z" * 5000
    line = b.headline(full, "f")
    assert len(line) < 4000
    h = json.loads(line)
    assert "roofline" in h and "cpu_baseline" in h and "value" in h
