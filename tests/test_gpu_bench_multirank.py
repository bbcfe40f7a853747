"""bench.py's N > 1 path driven for real on a 1-GPU box: two ranks share device 0 (ODO_BENCH_SHARE_GPU) and exchange poses
over gloo (ODO_BENCH_BACKEND) — the same sharding, schedule-based pose gather and max-over-ranks timing the 8-GPU RCCL run
uses. Uneven shards on purpose: 3 sequences over 2 ranks (BASELINE.json configs[3] in miniature: 11 over 8)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, world=2, timeout=900, no_extras=True, tmp_path=None):
    """Returns the full record (bench_details.json); the compact stdout line is checked here for the contract: ONE line, strict
    JSON, under 4 KB, carrying the headline keys (the 21 KB line came back unparsed from the driver)."""
    import tempfile
    details = os.path.join(tempfile.mkdtemp(prefix="odo_bench_"), "details.json")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, ODO_BENCH_SHARE_GPU="1", ODO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--cpu-frames", "0", "--details", details] + (["--no-extras"] if no_extras else []) + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]     # rank 0 prints ONE JSON line
    assert p.stdout.rstrip().splitlines()[-1] == lines[0]      # ... and it is the LAST line of stdout
    assert len(lines[0]) < 4096 and lines[0].isascii()
    line = json.loads(lines[0], parse_constant=lambda c: pytest.fail("non-strict JSON constant " + c))
    full = json.load(open(details))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in line, k
        if k not in ("config", "roofline"):
            assert line[k] == full[k], k
    assert line["config"]["workload"] and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    # N > 1: the exchange's evidence and the per-rank rates are in the compact line itself, scalars / flat lists only
    pg = line["pose_gather"]
    for k in ("backend", "rccl_ranks_seen", "distinct_devices", "complete"):
        assert k in pg and not isinstance(pg[k], (dict, list)), k
    assert pg["complete"] == full["pose_gather"]["complete"]
    assert len(line["per_rank"]["frames_per_s"]) == world
    full["_line"] = line
    return full


def test_bench_two_ranks_uneven_sequences():
    out = _run_bench(["--sequences", "3", "--steps", "12", "--warmup", "3", "--unique-frames", "16", "--gather-every", "5"])
    assert out["n_gpus"] == 2 and out["steps"] == 12 and out["scaling"] == "strong"
    assert out["config"]["sequences"] == 3 and out["config"]["frames_per_rank"] == [24, 12]
    g = out["pose_gather"]
    assert g["complete"] and g["rows_per_rank"] == [24, 12]
    assert g["collectives"] == 5                      # ceil(24 / 5) on EVERY rank, also the one that tracked 12 frames
    assert g["rank0_rows_match_tracked_poses"]
    assert out["value"] > 0 and abs(out["value"] - 36 / (out["ms_per_step"] * 1e-3 * 24)) < 1e-2 * out["value"]


def test_bench_two_ranks_default_weak_scaling():
    out = _run_bench(["--steps", "10", "--warmup", "2", "--unique-frames", "16", "--gather-every", "4"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["frames_per_rank"] == [10, 10]
    assert out["pose_gather"]["complete"] and out["pose_gather"]["rank0_rows_match_tracked_poses"]


def test_bench_two_ranks_carries_configs3_and_exchange_evidence():
    """The default N > 1 line: per-rank rates, the exchange's own evidence (ranks seen by a collective of the gather's backend,
    distinct devices), and configs[3] in miniature (3 sequences over the 2 ranks, batched rank 0) as an extra key."""
    out = _run_bench(["--steps", "8", "--warmup", "2", "--unique-frames", "12", "--gather-every", "4", "--configs3", "3"], no_extras=False)
    assert out["n_gpus"] == 2 and out["pose_gather"]["complete"]
    assert out["pose_gather"]["ranks_seen"] == 2 and out["pose_gather"]["backend"] == "gloo"
    assert out["pose_gather"]["distinct_devices"] == 1           # this test shares ONE device between the ranks on purpose
    assert len(out["per_rank"]["frames_per_s"]) == 2 and out["per_rank"]["slowest_over_fastest_seconds"] >= 1.0
    assert out["_line"]["configs3_sequences_3"]["frames_per_s"] == out["configs3_sequences_3"]["frames_per_s"]
    assert out["_line"]["configs3_sequences_3"]["gather_complete"] is True
    c3 = out["configs3_sequences_3"]
    assert c3["frames_per_rank"] == [16, 8] and c3["pose_gather"]["complete"] and c3["sequences_in_lock_step_on_rank0"] == 2
    assert c3["frames_per_s"] > 0
