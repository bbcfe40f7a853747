"""`python bench.py --gpus N` started plainly must start the N ranks itself (the flag used to be parsed and ignored —
a plain `bench.py --gpus 8` would have tracked on ONE GPU and printed n_gpus: 1). CPU test of the launcher through the
--launch-probe hook: the ranks do what a real rank does up to the point where it would touch the GPU (join the process group,
one all_reduce), so the spawn, the relay of rank 0's single JSON line and the exit codes are the real ones."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_plain_gpus_2_starts_two_ranks_and_relays_one_line():
    p, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-probe", "ok"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 3 and d["warmup"] == 1


def test_a_failing_rank_fails_the_run_and_prints_no_result():
    p, lines = _run(["--gpus", "2", "--launch-probe", "fail"])
    assert p.returncode != 0
    assert lines == []
    assert "a rank failed" in p.stderr


def test_a_line_with_another_n_gpus_is_refused():
    p, lines = _run(["--gpus", "2", "--launch-probe", "lie"])
    assert p.returncode != 0 and lines == []
    assert "refusing to print" in p.stderr


def test_gpus_flag_must_match_the_world_it_runs_in():
    p, lines = _run(["--gpus", "1", "--launch-probe", "ok"], extra_env=dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and lines == []
    assert "WORLD_SIZE = 2" in p.stderr


def test_single_gpu_default_does_not_spawn():
    p, lines = _run(["--launch-probe", "ok"])
    assert p.returncode == 0 and len(lines) == 1
    assert json.loads(lines[0])["n_gpus"] == 1
