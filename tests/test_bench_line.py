"""The driver parses the LAST stdout line of ``bench.py`` out of a bounded tail (~8.7 KB).  Round 5's line had grown
to 26 KB and the round went unmeasured; these tests pin the contract: whatever ``measure`` returns, ``main`` prints one
line under ``bench.MAX_LINE`` bytes that carries the contract's fields, ``roofline`` and ``cpu_baseline``."""

from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def fat_record(n_configs=80):
    """a full record shaped like round 5's, with more and longer entries than any real run has produced"""
    long = "x" * 700
    cfgs = {f"hot_case_{k}": {"config": long, "ms_device": 0.0123456789, "frac": 0.5123456789, "generated_kernels_us": {long[:60] + str(j): 1.0 for j in range(6)}}
            for k in range(n_configs)}
    for k in bench._BASELINE_CONFIGS:
        cfgs[k] = {"config": long, "frac": 0.4123456789123, "kernel_frac": 0.6123456789123, "kernel": long}
    return {
        "metric": "graph evals/sec (logp+grad, N=1e6 fp64)", "value": 5234.123456789, "unit": "graph evals/sec", "n_gpus": 1, "steps": 20, "warmup": 5,
        "warmup_effective": 800, "ms_per_step": 0.19123456789, "value_executor_level": 5400.123456, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": long, "mode": "hipGraph plan", "value_is": long, "function": {"compile_s": 1.0}, "parallelism": "replicas x1"},
        "roofline": {"bound": "hbm", "kernel": long, "achieved": 6221.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.7776543210987, "traffic": 1044553598.72,
                     "traffic_source": long, "detail": {"kernel_ms": 0.16712345678, "kernel_ms_top8": {long[:50] + str(j): 0.1 for j in range(8)}}},
        "cpu_baseline": {"value": 10.0880072, "unit": "graph evals/sec", "cores": 256, "kind": "reference-cvm", "sample": long, "ms_per_eval": 99.1276,
                         "parity_err_over_bound": 0.0794210265, "port": {"value": 3.0, "sample": long}},
        "configs": cfgs,
    }


def test_compact_is_bounded_and_complete():
    line = bench.compact(fat_record(), "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.MAX_LINE, len(text)
    for k in CONTRACT:
        assert k in line, k
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is not None
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "reference-cvm" and cpu["cores"] == 256 and cpu["value"] > 0 and cpu["sample"]
    assert set(line["configs"]) == set(bench._BASELINE_CONFIGS)  # one fraction each, no hot_* sweep
    assert all(isinstance(v, float) for v in line["configs"].values())
    assert "model" not in line["config"]


def test_compact_without_baseline_or_configs():
    rec = fat_record()
    rec["cpu_baseline"] = None
    rec["configs"] = None
    line = bench.compact(rec)
    assert line["cpu_baseline"] is None and line["configs"] == {}
    assert len(json.dumps(line)) < bench.MAX_LINE


def test_main_prints_one_short_last_line(monkeypatch, capsys, tmp_path):
    monkeypatch.setattr(bench, "measure", lambda args: fat_record(200))
    detail = tmp_path / "detail.json"
    bench.main(["--detail", str(detail)])
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1  # stdout carries the JSON line and nothing else
    last = lines[-1]
    assert len(last) < 6000
    rec = json.loads(last)
    assert rec["roofline"]["frac"] > 0 and rec["cpu_baseline"]["value"] > 0
    assert rec["steps"] == 20 and rec["warmup"] == 5 and rec["n_gpus"] == 1
    # the full record (hot_* sweep, per-kernel tables) is in the side file
    full = json.load(open(detail))
    assert len(full["configs"]) >= 200 and "detail" in full["roofline"]


def test_non_zero_ranks_print_nothing(monkeypatch, capsys):
    monkeypatch.setattr(bench, "measure", lambda args: None)
    bench.main([])
    assert capsys.readouterr().out == ""


def test_default_run_skips_the_sweep():
    args = bench.parse_args([])
    assert args.gpus == 1 and not args.hotpath
    assert args.steps * 0.2e-3 < 60  # default K at ~0.2 ms per evaluation: seconds, not minutes
