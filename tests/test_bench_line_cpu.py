"""bench.py's ONE stdout line: the compact contract line built from a full run record (no GPU: the builder is pure Python).

The driver parses the last stdout line; round 5's 24.7 KB line did not parse and the round went unmeasured.  These tests pin:
the line built from a real full record (profiles/r05_bench.json, the last record the old format printed) is < 4096 bytes,
strict JSON, has exactly the contract key set, scalars only below the second level, and carries the record's numbers.
"""
import json
import math
import os

import pytest

import bench
from conftest import ROOT


@pytest.fixture(scope="module")
def full():
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    rec = bench._finite(rec)
    rec["extras_file"] = bench.EXTRAS_FILE
    return rec


def _strict(text):
    return json.loads(text, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))


def test_line_is_small_strict_and_has_exactly_the_contract_keys(full):
    text = bench.dump_line(bench.contract_line(full))
    assert len(text.encode()) < bench.LINE_LIMIT == 4096 and "\n" not in text
    line = _strict(text)
    assert tuple(line) == bench.CONTRACT_KEYS
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line
    assert line["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["higher_is_better"] is True


def test_line_carries_the_records_numbers(full):
    line = bench.contract_line(full)
    assert line["value"] == pytest.approx(full["value"], rel=1e-8) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-6)
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4) and r["traffic"] == pytest.approx(full["roofline"]["traffic"], rel=1e-8)
    assert r["binding"] == "valu_fp64" and r["valu_fp64_frac"] == pytest.approx(full["roofline"]["valu_fp64"]["frac"], rel=1e-5)
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == full["cpu_baseline"]["cores"] and c["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-5)
    assert line["parity"]["index_exact"] is True and line["parity"]["max_abs_cost_err"] <= 1e-6
    assert line["legs"]["config4"]["ms_per_step"] == pytest.approx(full["config4"]["ms_per_step"], rel=1e-5) and line["legs"]["config4"]["parity_ok"] is True
    assert line["legs"]["closed_loop_FISS+"]["us_per_cycle"] == pytest.approx(full["closed_loop"]["FISS+"]["us_per_cycle"], rel=1e-5)


def test_line_has_scalars_only_below_the_second_level_and_short_strings(full):
    line = bench.contract_line(full)

    def walk(o, depth):
        if isinstance(o, dict):
            assert depth < 3, "the driver's reader keeps scalars of nested objects only"
            for v in o.values():
                walk(v, depth + 1)
        else:
            assert not isinstance(o, (list, tuple))
            if isinstance(o, str):
                assert len(o) <= 200
            if isinstance(o, float):
                assert math.isfinite(o)

    walk(line, 0)
    for k in ("config", "roofline", "cpu_baseline", "parity"):
        assert all(not isinstance(v, dict) for v in line[k].values()), k


def test_a_failed_parity_or_a_nan_cannot_hide(full):
    bad = json.loads(json.dumps(full))
    bad["config4"]["parity"]["stats_exact"] = False
    bad["config2"]["parity"]["max_abs_cost_err"] = 1e-3
    line = bench.contract_line(bad)
    assert line["legs"]["config4"]["parity_ok"] is False and line["legs"]["config2"]["parity_ok"] is False
    bad["parity"]["batches"][1]["index_exact"] = False
    assert bench.contract_line(bad)["parity"]["index_exact"] is False
    nan = bench._finite({"a": float("nan"), "b": [float("inf"), 1.0]})
    assert nan == {"a": None, "b": [None, 1.0]}
    with pytest.raises(ValueError):
        bench.dump_line({"x": float("nan")})


def test_an_oversized_line_is_refused(full):
    line = bench.contract_line(full)
    line["config"]["workload"] = "x" * 5000
    with pytest.raises(SystemExit):
        bench.dump_line(line)


def test_multi_gpu_record_without_the_single_gpu_legs(full):
    rec = {k: full[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                                "roofline")}
    rec.update(n_gpus=8, cpu_baseline=None, parity=None, extras_file="bench_extras.json")
    line = _strict(bench.dump_line(bench.contract_line(rec)))
    assert line["n_gpus"] == 8 and line["cpu_baseline"] is None and line["legs"] is None and line["parity"] is None


def test_emit_prints_one_line_and_writes_the_side_file(full, tmp_path, capsys, monkeypatch):
    monkeypatch.setenv("BENCH_EXTRAS_FILE", str(tmp_path / "x.json"))
    bench.emit(dict(full))
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and out.endswith("}\n")
    assert _strict(out)["extras_file"] == "x.json"
    side = json.load(open(tmp_path / "x.json"))
    assert side["config4"]["stage_ms"] and side["closed_loop"]["FOP"]["value"] > 0


def test_line_from_the_current_rounds_full_record():
    """The same on round 6's own full record (profiles/r06_bench_extras.json: the side file of the run behind profiles/r06_bench.json):
    the line bench.py printed equals the line built again from the record."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_extras.json")))
    printed = _strict(open(os.path.join(ROOT, "profiles", "r06_bench.json")).read())
    built = _strict(bench.dump_line(bench.contract_line(rec)))
    assert tuple(printed) == bench.CONTRACT_KEYS
    assert built == printed
    assert printed["legs"]["overlap"]["parity_ok"] is True and printed["legs"]["config4"]["ms_per_step"] < 0.21
    assert printed["roofline"]["traffic"] < 1.2 * printed["roofline"]["algorithmic_bytes_per_launch"]
