"""N > 1 path on CPU: world_size-2 gloo run of the sequence sharding and the batched pose gather (odometry_amd/dist.py).
The trackers themselves need a GPU; here each rank produces deterministic stand-in poses so the exchange is checked."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from odometry_amd.dist import PoseGatherer, shard


def fake_pose(rank, frame):
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [rank + 0.25, frame * 0.5, -frame - rank * 100.0]
    T[0, 1] = 0.001 * frame
    return T


def _worker(rank, world, port, n_frames, every, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = PoseGatherer(world, every)
    for f in range(n_frames):
        g.push(fake_pose(rank, f))
    g.flush()
    dist.barrier()
    q.put((rank, [g.poses(r) for r in range(world)]))
    dist.destroy_process_group()


def test_shard_is_disjoint_and_covering():
    for n in (1, 8, 11):
        for world in (1, 2, 4, 8):
            got = sorted(i for r in range(world) for i in shard(n, r, world))
            assert got == list(range(n))
    assert shard(11, 3, 8) == [3]          # config 4: 11 sequences over 8 GPUs
    assert shard(11, 2, 8) == [2, 10]


def test_pose_gather_world2_gloo():
    world, n_frames, every = 2, 19, 8       # 19 is not a multiple of 8: the tail flush is exercised
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, every, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for viewer in range(world):
        for r in range(world):
            got = res[viewer][r]
            assert got.shape == (n_frames, 3, 4)
            for f in range(n_frames):
                assert np.array_equal(got[f], fake_pose(r, f)[:3, :])
