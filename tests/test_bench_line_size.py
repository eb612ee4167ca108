"""The ONE stdout line of bench.py stays small enough for the driver to parse (round-5 review: 25.9 KB of counter tables and
notes in the line left BENCH_r05.json.parsed = null).  No GPU: the line is rebuilt from a committed full record of the DEFAULT
workload set (200^3 headline + the secondary workloads + CPU baseline + roofline with counters)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RECORDS = ["r06p_bench_detail.json", "r05g_bench_line.json", "r04zq_poisson200_bench_line.json", "r04y_poisson200_exchange_selftest_share_as_world8_bench_line.json"]


@pytest.mark.parametrize("rec", RECORDS)
def test_compact_line_fits_and_keeps_the_contract(rec):
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", rec)))
    txt = bench.compact_line(full)
    assert len(txt) < bench.LINE_LIMIT <= 4096 and "\n" not in txt
    d = json.loads(txt, parse_constant=lambda c: pytest.fail("non-finite constant in the line: " + c))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-9) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-9)
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "mfma_utilisation", "kernel"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and abs(r["frac"] - full["roofline"]["frac"]) < 1e-6
    if full.get("cpu_baseline"):
        for k in ("value", "unit", "cores", "host_cores", "kind", "sample"):
            assert k in d["cpu_baseline"], k
    # no prose: every string in the line is a short identifier
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(x) for x in strings(d)) <= 160
    if full.get("secondary"):
        assert len(d["secondary"]) == len(full["secondary"])


def test_compact_line_survives_non_finite_and_errors():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", RECORDS[0])))
    full["residual_2norm"] = float("nan")
    full["roofline"]["traffic"] = float("inf")
    full["secondary"][0] = {"workload": "x" * 500, "error": "boom " * 100}
    full["error"] = "deadline " * 200
    d = json.loads(bench.compact_line(full))
    assert "residual_2norm" not in d or d["residual_2norm"] is None
    assert d["roofline"]["traffic"] is None and len(d["error"]) <= 160


def test_the_committed_line_is_what_the_full_record_compacts_to():
    """profiles/r06p_bench_line.json is the stdout of the run whose full record is r06p_bench_detail.json"""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r06p_bench_detail.json")))
    line = open(os.path.join(ROOT, "profiles", "r06p_bench_line.json")).read().strip()
    assert len(line) < 4096 and json.loads(line) == json.loads(bench.compact_line(full))
    assert len(json.loads(line)["secondary"]) == 5
