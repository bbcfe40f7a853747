"""N > 1 path on CPU: world_size-2 gloo run of the sequence sharding and the batched pose gather (odometry_amd/dist.py).
The trackers themselves need a GPU; here each rank produces deterministic stand-in poses so the exchange is checked."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from odometry_amd.dist import PoseGatherer, frames_per_rank, shard


def fake_pose(rank, frame):
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [rank + 0.25, frame * 0.5, -frame - rank * 100.0]
    T[0, 1] = 0.001 * frame
    return T


def _worker(rank, world, port, n_frames, every, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = PoseGatherer(world, every, n_local_frames=n_frames)
    for f in range(n_frames):
        g.push(fake_pose(rank, f))
    g.flush()
    dist.barrier()
    q.put((rank, [g.poses(r) for r in range(world)]))
    dist.destroy_process_group()


def _worker_uneven(rank, world, port, n_sequences, frames, every, agree, q):
    """BASELINE.json configs[3] in miniature: n_sequences sequences dealt over `world` ranks, so the ranks push different
    numbers of rows; the collective schedule must not depend on that."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per_rank = frames_per_rank(n_sequences, frames, world)
    g = PoseGatherer(world, every, n_local_frames=per_rank[rank], n_max_frames=None if agree else max(per_rank))
    for sid in shard(n_sequences, rank, world):
        for f in range(frames):
            g.push(fake_pose(sid, f), seq_id=sid, frame_id=f)
    g.flush()
    dist.barrier()
    q.put((rank, g.issued, [g.rows(r) for r in range(world)]))
    dist.destroy_process_group()


def _worker_mixed(rank, world, port, q):
    """One rank passes n_max_frames, the other leaves it to the constructor: the agreement collective runs on BOTH (it used to run
    only on ranks without n_max_frames and hang). Ids beyond 2^24 survive the float32 row."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local = [5, 3][rank]
    g = PoseGatherer(world, 4, n_local_frames=n_local, n_max_frames=5 if rank == 0 else None)
    for f in range(n_local):
        g.push(fake_pose(rank, f), seq_id=(1 << 24) + 1 + rank, frame_id=(1 << 30) + f)
    g.flush()
    dist.barrier()
    err = None
    try:
        PoseGatherer(world, 4)          # legacy mode is refused when world > 1
    except ValueError as e:
        err = str(e)
    q.put((rank, g.n_max, [g.ids(r) for r in range(world)], err))
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_frames_per_rank_matches_shard():
    assert frames_per_rank(11, 200, 8) == [400, 400, 400, 200, 200, 200, 200, 200]   # configs[3]
    assert frames_per_rank(3, 7, 2) == [14, 7]
    assert frames_per_rank(1, 5, 2) == [5, 0]                                      # a rank without a sequence


import pytest


@pytest.mark.parametrize("agree", [False, True])    # schedule from the sharding rule / agreed by one all_reduce(MAX)
@pytest.mark.parametrize("n_sequences,frames,every", [(3, 7, 4), (1, 5, 8), (5, 6, 3)])
def test_pose_gather_uneven_shards_world2_gloo(n_sequences, frames, every, agree):
    """Ranks track different numbers of frames (3 sequences over 2 ranks: 14 vs 7 rows; one rank with nothing at all): every
    rank issues the same number of fixed-size collectives, nothing hangs, every row arrives everywhere with its tags."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, n_sequences, frames, every, agree, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, issued, rows = q.get(timeout=120)
        res[rank] = (issued, rows)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per_rank = frames_per_rank(n_sequences, frames, world)
    n_coll = (max(per_rank) + every - 1) // every
    for viewer in range(world):
        issued, rows = res[viewer]
        assert issued == n_coll                          # the schedule, not the local frame count
        for r in range(world):
            got = rows[r]
            assert got.shape == (per_rank[r], 14)
            k = 0
            for sid in shard(n_sequences, r, world):
                for f in range(frames):
                    assert tuple(got[k, :2].view(np.int32)) == (sid, f)     # ids travel as int32 bit patterns
                    assert np.array_equal(got[k, 2:].reshape(3, 4), fake_pose(sid, f)[:3, :])
                    k += 1


def test_pose_gather_mixed_n_max_and_large_ids_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_mixed, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, n_max, ids, err = q.get(timeout=120)
        res[rank] = (n_max, ids, err)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for viewer in range(world):
        n_max, ids, err = res[viewer]
        assert n_max == 5 and err and "n_local_frames is required" in err
        for r, n_local in enumerate([5, 3]):
            assert ids[r].shape == (n_local, 2)
            assert ids[r][:, 0].tolist() == [(1 << 24) + 1 + r] * n_local            # not representable in float32
            assert ids[r][:, 1].tolist() == [(1 << 30) + f for f in range(n_local)]


def _worker_disagree(rank, world, port, q):
    """The ranks state DIFFERENT n_max_frames: every rank must raise — none may carry on into a collective the other never joins."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    err = None
    try:
        PoseGatherer(world, 4, n_local_frames=3, n_max_frames=[5, 7][rank])
    except ValueError as e:
        err = str(e)
    err2 = None
    try:
        PoseGatherer(world, 4, n_local_frames=[9, 3][rank], n_max_frames=5)   # smaller than the largest shard: an error everywhere too
    except ValueError as e:
        err2 = str(e)
    dist.barrier()        # both ranks are still in step
    q.put((rank, err, err2))
    dist.destroy_process_group()


def test_disagreeing_n_max_frames_raises_on_every_rank():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_disagree, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, err, err2 = q.get(timeout=120)
        res[rank] = (err, err2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r][0] and "different n_max_frames" in res[r][0]
        assert res[r][1] and "different n_max_frames" in res[r][1]
