"""bench.py's result line (VERDICT r05 task 1: the 21 KB single line of round 5 came back from the driver unparsed). The LAST stdout
line is a compact contract line (< 4 KB, ASCII, strict JSON); the full record goes to bench_details.json. CPU test of the formatter on
the real record of round 5 (profiles/r05_bench_steps20.json, 21 KB) and on hostile variants of it."""
import copy
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _canned():
    with open(os.path.join(ROOT, "profiles", "r05_bench_steps20.json")) as f:
        return json.load(f)


def _strict(text):
    def bad(c):
        raise AssertionError("non-strict JSON constant " + c)
    return json.loads(text, parse_constant=bad)


def test_compact_line_of_the_round5_record_fits_and_carries_the_contract():
    out = _canned()
    assert len(json.dumps(out)) > 20000          # the record that did not parse
    text = bench.format_result_line(out)
    assert "\n" not in text and text.isascii() and len(text) < bench.COMPACT_MAX_BYTES
    d = _strict(text)
    for k in CONTRACT:
        assert k in d, k
    assert d["value"] == out["value"] and d["ms_per_step"] == out["ms_per_step"] and d["dtype"] == "f32"
    assert d["config"]["workload"].startswith("configs[1]") and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["achieved"] == out["roofline"]["achieved"] and r["frac"] == out["roofline"]["frac"]
    assert r["traffic"] == out["roofline"]["traffic"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    c = d["cpu_baseline"]
    assert c["value"] == out["cpu_baseline"]["value"] and c["cores"] == 1 and c["kind"] == "port" and c["sample"]
    assert d["pose_max_abs_delta_vs_oracle"] == 0.0 and d["speedup_vs_cpu"] == out["speedup_vs_cpu"]
    # one scalar per side leg
    assert d["dense_1080p_frac"] == out["roofline_dense_1080p"]["frac"]
    assert d["scan_us"] == out["disparity_1241x376"]["full_range"]["scan_us"]
    assert d["shim_cvmat_load_per_frame_fps"] == out["shim_path_load_per_frame"]["cvmat"]["frames_per_s"]
    assert d["batched_s8_fps"] == out["batched_sequences"]["batched"][-1]["frames_per_s"]
    assert all(not isinstance(v, (dict, list)) for k, v in d.items() if k not in ("config", "roofline", "cpu_baseline"))


def test_compact_line_survives_missing_legs_nan_and_failed_legs():
    out = _canned()
    for k in ("roofline_dense_1080p", "disparity_1241x376", "batched_sequences", "shim_path", "cpu_baseline", "step_us"):
        out.pop(k)
    out["roofline"]["achieved"] = float("nan")
    out["roofline"]["frac"] = float("inf")
    out["roofline_dense_1080p_error"] = "RuntimeError: boom"
    d = _strict(bench.format_result_line(out))
    assert d["roofline"]["achieved"] is None and d["roofline"]["frac"] is None
    assert "cpu_baseline" not in d and d["failed_legs"] == ["roofline_dense_1080p_error"]
    assert d["value"] == out["value"]


def test_compact_line_of_an_n_gpu_record_names_the_exchange():
    out = _canned()
    out.update(n_gpus=8, pose_gather=dict(rows_per_rank=[20] * 8, complete=True, collectives=1, rows_per_collective=32, backend="nccl",
                                          rccl_ranks_seen=8, ranks_seen=8, distinct_devices=8, rank0_rows_match_tracked_poses=True),
               per_rank=dict(frames=[20] * 8, seconds=[0.0055] * 8, frames_per_s=[3600.0] * 8, slowest_over_fastest_seconds=1.0),
               configs3_sequences_11=dict(frames_per_s=20000.0, scaling="strong", pose_gather=dict(complete=True), what="x" * 500))
    text = bench.format_result_line(out)
    assert len(text) < bench.COMPACT_MAX_BYTES
    d = _strict(text)
    assert d["pose_gather"] == dict(backend="nccl", rccl_ranks_seen=8, distinct_devices=8, complete=True, collectives=1,
                                    rank0_rows_match_tracked_poses=True)
    assert d["per_rank"]["frames_per_s"] == [3600.0] * 8
    assert d["configs3_sequences_11"] == dict(frames_per_s=20000.0, scaling="strong", gather_complete=True)


def test_an_oversized_record_is_trimmed_to_the_contract_keys_not_dropped():
    out = _canned()
    for i in range(400):
        out["batched_sequences"]["batched"].append(dict(sequences=100 + i, frames_per_s=1.0 + i))
    text = bench.format_result_line(out)
    assert len(text) < bench.COMPACT_MAX_BYTES
    d = _strict(text)
    assert d["truncated"] is True
    for k in CONTRACT:
        assert k in d, k


def test_emit_result_writes_details_and_prints_the_compact_line_last(tmp_path):
    out = _canned()
    path = tmp_path / "details.json"
    so, se = io.StringIO(), io.StringIO()
    with redirect_stdout(so), redirect_stderr(se):
        print("a leg's chatter")
        bench.emit_result(out, str(path))
    lines = so.getvalue().rstrip().splitlines()
    assert _strict(lines[-1])["value"] == out["value"] and len(lines[-1]) < bench.COMPACT_MAX_BYTES
    assert sum(ln.startswith("{") for ln in lines) == 1          # ONE JSON line on stdout
    full = json.load(open(path))
    assert full["roofline"]["lm_fine_kernel"] == out["roofline"]["lm_fine_kernel"]      # nothing lost: the full record is on file
    assert se.getvalue().startswith("[bench details] {")
