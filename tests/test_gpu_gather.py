"""odo_gather_* — the pose exchange of the multi-GPU path behind the C ABI (RCCL, dlopen'ed). A 1-GPU box can only run a
communicator of one rank (RCCL refuses two ranks on one device): the schedule, the padding, the row layout and the RCCL calls are
exercised at world size 1; the N > 1 schedule itself is the one odometry_amd/dist.py runs in tests/test_distributed_gloo.py."""
import ctypes as C

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
        assert rows[k, 0] == 3 + (k % 2) and rows[k, 1] == k
        assert np.array_equal(rows[k, 2:].reshape(3, 4), poses[k][:3, :])
    assert lib.odo_gather_rows(g, 1, C.byref(rows_p), C.byref(n)) != 0   # no such rank
    L.check(lib.odo_gather_destroy(g), "odo_gather_destroy")
