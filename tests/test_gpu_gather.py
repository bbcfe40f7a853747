"""odo_gather_* — the pose exchange of the multi-GPU path behind the C ABI (RCCL, dlopen'ed). A 1-GPU box can only run a
communicator of one rank (RCCL refuses two ranks on one device): the schedule, the padding, the row layout and the RCCL calls are
exercised at world size 1; the two-rank tests below (real RCCL over two devices, the C ABI and bench.py's torch `nccl` route) run
wherever torch.cuda.device_count() >= 2 and are skipped with that reason elsewhere."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("every,n_local,n_max", [(4, 10, 10), (4, 10, 17), (8, 0, 5), (3, 9, 9)])
def test_pose_gather_world_one(every, n_local, n_max):
    from odometry_amd import _lib as L
    lib = L.load()
    uid = (C.c_ubyte * 128)()
    L.check(lib.odo_gather_unique_id(uid), "odo_gather_unique_id")
    g = C.c_void_p()
    L.check(lib.odo_gather_create(0, 1, 0, uid, every, n_local, n_max, C.byref(g)), "odo_gather_create")
    assert lib.odo_gather_ranks(g) == 1                        # ncclCommCount of the communicator
    rng = np.random.default_rng(every * 100 + n_local)
    poses = []
    for k in range(n_local):
        T = np.eye(4, dtype=np.float32)
        T[:3, :] = rng.standard_normal((3, 4)).astype(np.float32)
        poses.append(T)
        col = np.ascontiguousarray(T.T).reshape(-1)          # column-major, as the tracker returns it
        L.check(lib.odo_gather_push(g, 3 + (k % 2), k, col.ctypes.data_as(C.POINTER(C.c_float))), "odo_gather_push")
        assert lib.odo_gather_issued(g) == (k + 1) // every    # a collective per `every` rows, issued without waiting
    # one more row than announced is refused
    col = np.zeros(16, np.float32)
    assert lib.odo_gather_push(g, 0, 0, col.ctypes.data_as(C.POINTER(C.c_float))) != 0
    L.check(lib.odo_gather_flush(g), "odo_gather_flush")
    assert lib.odo_gather_issued(g) == -(-n_max // every)      # the schedule depends on n_max alone
    rows_p, n = C.POINTER(C.c_float)(), C.c_int(0)
    L.check(lib.odo_gather_rows(g, 0, C.byref(rows_p), C.byref(n)), "odo_gather_rows")
    assert n.value == n_local
    rows = np.ctypeslib.as_array(rows_p, shape=(max(n.value, 1), 14))[:n.value].copy() if n.value else np.zeros((0, 14), np.float32)
    for k in range(n_local):
        assert tuple(rows[k, :2].view(np.int32)) == (3 + (k % 2), k)     # ids are int32 bit patterns in the float row
        assert np.array_equal(rows[k, 2:].reshape(3, 4), poses[k][:3, :])
    assert lib.odo_gather_rows(g, 1, C.byref(rows_p), C.byref(n)) != 0   # no such rank
    L.check(lib.odo_gather_destroy(g), "odo_gather_destroy")


def _n_devices():
    import torch
    return torch.cuda.device_count()      # counting devices does not initialise the GPU in this process


_RANK_SCRIPT = r"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, {root!r})
from odometry_amd import _lib as L
rank, world, idfile, every = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
n_local = [7, 3][rank]
lib = L.load()
uid = (C.c_ubyte * 128).from_buffer_copy(open(idfile, "rb").read())
g = C.c_void_p()
L.check(lib.odo_gather_create(rank, world, rank, uid, every, n_local, 7, C.byref(g)), "odo_gather_create")
assert lib.odo_gather_ranks(g) == world
for k in range(n_local):
    T = np.eye(4, dtype=np.float32); T[:3, 3] = [rank, k, -k]
    L.check(lib.odo_gather_push(g, 10 + rank, k, np.ascontiguousarray(T.T).reshape(-1).ctypes.data_as(C.POINTER(C.c_float))), "push")
L.check(lib.odo_gather_flush(g), "flush")
assert lib.odo_gather_issued(g) == -(-7 // every)
for r, n_want in enumerate([7, 3]):
    rows_p, n = C.POINTER(C.c_float)(), C.c_int(0)
    L.check(lib.odo_gather_rows(g, r, C.byref(rows_p), C.byref(n)), "rows")
    assert n.value == n_want, (r, n.value)
    rows = np.ctypeslib.as_array(rows_p, shape=(n.value, 14)).copy()
    assert rows[:, :2].view(np.int32).tolist() == [[10 + r, k] for k in range(n_want)]
    assert rows[:, 5].tolist() == [float(r)] * n_want and rows[:, 9].tolist() == [float(k) for k in range(n_want)]
L.check(lib.odo_gather_destroy(g), "destroy")
print("RANK_OK", rank)
"""


def test_two_ranks_rccl_through_the_c_abi(tmp_path):
    """odo_gather_* at world size 2: two processes, one device each, ncclCommInitRank from a shared ncclUniqueId, uneven shards
    (7 vs 3 rows), every rank receives every row."""
    if _n_devices() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device (this box has %d)" % _n_devices())
    from odometry_amd import _lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    uid = (C.c_ubyte * 128)()
    L.check(L.load().odo_gather_unique_id(uid), "odo_gather_unique_id")
    idfile = tmp_path / "uid.bin"
    idfile.write_bytes(bytes(uid))
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=root))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(idfile), "4"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0 and f"RANK_OK {r}" in out, err[-2000:]


def test_two_ranks_bench_self_launch_over_nccl():
    """`python bench.py --gpus 2` launched plainly: the parent starts the two ranks itself, they exchange poses over the torch
    `nccl` backend (RCCL), rank 0's one JSON line comes back with n_gpus == 2 and the exchange checked."""
    if _n_devices() < 2:
        pytest.skip("needs two GPUs: the nccl backend wants one device per rank (this box has %d)" % _n_devices())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
                        "--unique-frames", "16", "--gather-every", "5", "--no-extras", "--cpu-frames", "0"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["pose_gather"]["complete"] and out["pose_gather"]["backend"] == "nccl"
    assert out["pose_gather"]["rccl_ranks_seen"] == 2 and out["pose_gather"]["distinct_devices"] == 2
