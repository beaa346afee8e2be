"""bench.py's stdout contract: ONE line, strict JSON, < 6 KB -- the driver keeps ~8 KB of stdout tail and the round-5
line (20.7 KB) was lost to it. run() is driven here on the CPU emulation of the kernels (tests/emu) at a small scale, with the
in-run parity gate against the oracle switched on, then the record is blown up to its worst case (live kernel table, sharded
sections, long strings) and compact_line must still fit."""
import copy
import glob
import json
import os
import subprocess

import pytest

import bench
from automerge_classic_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libam355_emu.so")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def strict(text):
    def bad(x):
        raise ValueError("non-finite constant in the line: " + x)
    return json.loads(text, parse_constant=bad)


@pytest.fixture(scope="module")
def record(tmp_path_factory):
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    eng = engine.Engine(0, EMU_LIB)
    path = str(tmp_path_factory.mktemp("bench") / "bench_detail.json")
    args = bench.parse_args(["--steps", "2", "--warmup", "1", "--prewarm", "0", "--scale", "0.02", "--subline-scale", "0.004", "--no-live-trace", "--detail", path])
    saved = bench.js_end_to_end, bench.reference_js_baseline, bench.sharded_js, bench.js_apply_latency
    bench.js_apply_latency = lambda log, calls=60: {"ms_per_call": 0.3, "first_call_ms": 1.5, "calls": 60, "ms": {}, "timed_region": "stub"}
    bench.sharded_js = lambda *a, **k: {"workload": "stub", "n_gpus": 1, "engine_ms_per_step": 1.0, "ops_per_s": 1.0e9, "per_rank_ms": [], "parity": "patch text sha256 == the unsharded engine's"}
    bench.js_end_to_end = lambda log: {"t_e2e_ms": 12.5, "t_e2e_ops_per_s": 8.0e7, "t_replay_ms_through_node": 1.0, "ms": {}, "timed_region": "stub"}
    bench.reference_js_baseline = lambda *a, **k: None   # (the GPU box's situation: no reference tree -> the C port leg + the recorded figure)
    try:
        detail = bench.run(args, eng, 0, 1, None, None, lambda: None, lambda: None)
    finally:
        bench.js_end_to_end, bench.reference_js_baseline, bench.sharded_js, bench.js_apply_latency = saved
        eng.close()
    return detail, path


def test_line_is_one_strict_json_line_under_6k(record, capsys):
    detail, path = record
    line = bench.emit(copy.deepcopy(detail), path)
    out = capsys.readouterr().out.strip().splitlines()
    assert out[-1] == line and len(out) == 1          # stdout: the line and nothing else
    assert len(line) < 6144
    d = strict(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["config"]["parity_checked_in_run"] is True and len(d["config"]["parity"]) <= 200
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    names = [r["name"] for r in d["workloads"]]
    assert any(n.startswith("c5_doc_mixed") for n in names) and any("shuffled" in n for n in names)
    assert [b[0] for b in d["apply_changes_ms"]] == [b["batch_changes"] for b in detail["apply_changes"]["batches"]]
    # the whole record went to the side file, strict JSON as well
    with open(path) as f:
        full = strict(f.read())
    assert "phases_ms" in full and "sharding_model" in full and len(full["workloads"]) == len(d["workloads"])
    assert d["sharded_js"]["n_gpus"] == 1 and "parity" in d["sharded_js"]
    assert d["js_apply_changes_ms_per_call"] == 0.3


def test_worst_case_record_still_fits(record):
    detail, _ = record
    d = copy.deepcopy(detail)
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_table.json")))
    with open(tables[-1]) as f:
        t = json.load(f)
    d["roofline"]["kernels"] = t["kernels"] * 3
    d["roofline"]["traffic"] = 506608694
    d["roofline"]["kernels_live"] = True
    d["value"] = float("nan")                      # (a NaN must not reach the line as a bare NaN token)
    d["config"]["workload"] = "w" * 2000
    d["config"]["parity"] = "p" * 2000
    d["cpu_baseline"]["sample"] = "s" * 5000
    d["cpu_baseline"]["reference_js_recorded"] = {"value": 27903.0, "file": "profiles/r06_reference_js_baseline.json", "recorded": "2026-09-30", "sample": "x" * 3000}
    big = {"n_gpus": 8, "scaling": "strong", "ops_per_s": 1.0e9, "ms_per_step": 1.0, "single_gpu_ms_per_step": 1.2, "speedup_vs_single_gpu": 1.2, "parity": "q" * 500}
    d["sharded"], d["sharded_c5"] = big, dict(big)
    d["workloads"] = d["workloads"] * 4
    line = bench.emit(d, None)
    assert len(line) < 6144
    out = strict(line)
    assert out["value"] is None and "dominant_kernel" in out["roofline"] and out["roofline"]["traffic"] == 506608694


def test_parity_gate_refuses_to_time_a_wrong_patch(monkeypatch):
    """A patch that differs from the oracle's ends the run before any timing."""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    eng = engine.Engine(0, EMU_LIB)
    monkeypatch.setattr(bench, "oracle_patch_sha256", lambda log: "0" * 64)
    args = bench.parse_args(["--steps", "1", "--warmup", "0", "--prewarm", "0", "--scale", "0.02", "--no-sublines", "--no-live-trace"])
    try:
        with pytest.raises(SystemExit, match="PARITY GATE FAILED"):
            bench.run(args, eng, 0, 1, None, None, lambda: None, lambda: None)
    finally:
        eng.close()
